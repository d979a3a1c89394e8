#!/usr/bin/env python3
"""From the kernel traces of tools/prof_all.sh: per workload the five (kernel, grid) pairs with the
most device time and their average duration per dispatch (the stats files average a kernel over all
of its launches, the small bins included).
   python3 tools/dominant_kernels.py gpurun_out/prof_final > profiles/<tag>_dominant_kernels.txt"""
import collections, csv, glob, os, sys

src = sys.argv[1]
for wdir in sorted(glob.glob(os.path.join(src, "*"))):
    traces = glob.glob(f"{wdir}/trace/**/*kernel_trace.csv", recursive=True)
    if not traces:
        continue
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(traces[0])):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        by[(name, int(r["Grid_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    print(f"== {os.path.basename(wdir)}: the five (kernel, grid) pairs with the most device time; average duration per dispatch")
    for (name, grid), d in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:5]:
        print(f"{name:40s} grid {grid:11d}  dispatches {len(d):3d}  avg {sum(d) / len(d):9.4f} ms  min {min(d):9.4f}  max {max(d):9.4f}")
