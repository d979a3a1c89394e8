#!/usr/bin/env python3
"""Registers, scratch and LDS of the kernels in spgemm_amd/csrc/spgemm_capi.s (make -C spgemm_amd/csrc asm)
whose mangled name contains one of the given fragments."""
import re, sys
s = open(sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".s") else "spgemm_amd/csrc/spgemm_capi.s").read()
frags = [a for a in sys.argv[1:] if not a.endswith(".s")]
for b in s.split("  - .agpr_count")[1:]:
    nm = re.search(r"\.name:\s+(\S+)", b)
    if not nm or (frags and not any(f in nm.group(1) for f in frags)):
        continue
    g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, b).group(1)
    print(nm.group(1)[:64], "vgpr", g("vgpr_count"), "sgpr", g("sgpr_count"), "scratch", g("private_segment_fixed_size"))
