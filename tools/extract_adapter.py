#!/usr/bin/env python3
"""Writes the adapter that the integration guide (DOC below) documents (the text between its [adapter:begin] and
[adapter:end] lines) into a header, so that the tests compile and run exactly what the document
shows.  usage: extract_adapter.py <out.hpp>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOC = "INTEGRATION.md"


def adapter_text() -> str:
    doc = open(os.path.join(ROOT, DOC)).read()
    a = doc.index("// [adapter:begin]")
    b = doc.index("// [adapter:end]")
    return doc[a:b] + "// [adapter:end]\n"


if __name__ == "__main__":
    with open(sys.argv[1], "w") as f:
        f.write(adapter_text())
