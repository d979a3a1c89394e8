#!/usr/bin/env python3
"""Writes the operands and engine options of every case of tests/test_gpu_fuzz.py to one binary file
(argv[1]) for tools/asan_replay/replay.cpp.  Per case: u64 n_options, then per option a 32-byte
zero-padded name and an i64 value; then A and B, each as u64 rows, cols, nnz, row_ptr u64[rows + 1],
col_idx u32[nnz], val f64[nnz]."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as tf  # noqa: E402


def put(f, M):
    np.array([M.rows, M.cols, M.nnz], dtype=np.uint64).tofile(f)
    np.ascontiguousarray(M.row_ptr, dtype=np.uint64).tofile(f)
    np.ascontiguousarray(M.col_idx, dtype=np.uint32).tofile(f)
    np.ascontiguousarray(M.val, dtype=np.float64).tofile(f)


with open(sys.argv[1], "wb") as f:
    for seed in range(tf.N_CASES):
        A, B, opts = tf.make_case(seed)
        np.array([len(opts)], dtype=np.uint64).tofile(f)
        for name, value in opts.items():
            f.write(name.encode().ljust(32, b"\0"))
            np.array([value], dtype=np.int64).tofile(f)
        put(f, A)
        put(f, B)
print(f"{tf.N_CASES} cases written to {sys.argv[1]}")
