"""The adapter that the integration guide shows a maintainer of the reference (a Drop* operator whose B blocks
stay on the GPU) is compiled from the document's own text against stand-ins for the reference's
types (tests/integration/ref_stubs.hpp), and run on the GPU against the reference's triple loop."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import extract_adapter  # noqa: E402

HARNESS = os.path.join(ROOT, "tests", "integration", "adapter_harness")


def test_documented_adapter_compiles_with_gxx_only(tmp_path):
    """g++ alone (no hipcc, no HIP headers): the adapter needs nothing but include/spgemm_amd.h."""
    (tmp_path / "adapter_extracted.hpp").write_text(extract_adapter.adapter_text())
    (tmp_path / "tu.cpp").write_text('#include "ref_stubs.hpp"\n#include "adapter_extracted.hpp"\n'
                                     "int main() { return sizeof(mults::HipBlocks) ? 0 : 1; }\n")
    p = subprocess.run(["g++", "-std=c++20", "-Wall", "-Wextra", "-Werror", "-c", "-I", os.path.join(ROOT, "include"),
                        "-I", os.path.join(ROOT, "tests", "integration"), "-I", str(tmp_path), "-o", str(tmp_path / "tu.o"),
                        str(tmp_path / "tu.cpp")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-4000:]


def test_prebuilt_harness_holds_the_documented_text():
    """The harness binary the GPU test runs is built (by __graft_entry__.build) from the header
    extracted next to it; that header must be the document's current text."""
    path = os.path.join(ROOT, "tests", "integration", "adapter_extracted.hpp")
    if not os.path.exists(path):
        pytest.skip("not built yet: python -c 'import __graft_entry__ as g; g.build()'")
    assert open(path).read() == extract_adapter.adapter_text(), "the documented adapter changed: rebuild (build())"


@pytest.mark.gpu
def test_documented_adapter_gives_the_reference_loops_bits():
    assert os.path.exists(HARNESS), "build it: python -c 'import __graft_entry__ as g; g.build()'"
    p = subprocess.run([HARNESS], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert p.stdout.strip().endswith("OK") and p.stdout.count("same bits") == 6, p.stdout
