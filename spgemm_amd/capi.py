"""ctypes binding of the C-ABI library (include/spgemm_amd.h).

This module only marshals pointers; every computation happens in libspgemm_amd.so on a HIP
device.  There is no CPU fallback: loading fails loudly when the library is missing, and every
compute call raises SpgemmError when no GPU is usable.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPGEMM_AMD_LIB: path of an alternative build of the same library (kernel tuning experiments)
LIB_PATH = os.environ.get("SPGEMM_AMD_LIB") or os.path.join(_HERE, "libspgemm_amd.so")

# every symbol include/spgemm_amd.h declares
EXPORTED_SYMBOLS = [
    "spgemm_version", "spgemm_last_error", "spgemm_device_count",
    "spgemm_create", "spgemm_destroy", "spgemm_ctx_device", "spgemm_set_stream", "spgemm_synchronize", "spgemm_set_option", "spgemm_device_property",
    "spgemm_mat_from_host", "spgemm_mat_wrap_device", "spgemm_mat_invalidate", "spgemm_mat_info", "spgemm_mat_device_ptrs",
    "spgemm_mat_to_host", "spgemm_mat_rows_to_host", "spgemm_mat_check", "spgemm_mat_free",
    "spgemm_multiply", "spgemm_symbolic", "spgemm_numeric", "spgemm_get_stats", "spgemm_get_launches",
    "spgemm_section_bitmap",
    "spgemm_partition_naive", "spgemm_partition_balanced", "spgemm_row_nprod",
    "spgemm_shuffle_random", "spgemm_shuffle_iterative", "spgemm_mat_select", "spgemm_mat_relabel_columns",
    "spgemm_section_nnz", "spgemm_pack_sections", "spgemm_assemble_row_ptr", "spgemm_assemble_row_ptr_noalloc",
    "spgemm_buffer_alloc", "spgemm_buffer_free", "spgemm_buffer_copy", "spgemm_buffer_upload", "spgemm_upload_sections",
]


class SpgemmError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"spgemm_amd error {code}: {msg}")
        self.code = code


class Partition(ctypes.Structure):
    _fields_ = [("start_row", ctypes.c_uint32), ("end_row", ctypes.c_uint32),
                ("start_col", ctypes.c_uint32), ("end_col", ctypes.c_uint32)]

    def as_tuple(self) -> Tuple[int, int, int, int]:
        return (self.start_row, self.end_row, self.start_col, self.end_col)


class Stats(ctypes.Structure):
    _fields_ = [("rows_a", ctypes.c_uint64), ("nnz_a", ctypes.c_uint64), ("nnz_b", ctypes.c_uint64),
                ("nprod", ctypes.c_uint64), ("nnz_structural", ctypes.c_uint64), ("nnz_c", ctypes.c_uint64),
                ("bytes_algorithmic", ctypes.c_uint64), ("temp_bytes", ctypes.c_uint64),
                ("rows_window", ctypes.c_uint32), ("rows_hash", ctypes.c_uint32),
                ("rows_global", ctypes.c_uint32), ("rows_empty", ctypes.c_uint32),
                ("ms_analyze", ctypes.c_float), ("ms_symbolic", ctypes.c_float), ("ms_scan", ctypes.c_float),
                ("ms_numeric", ctypes.c_float), ("ms_compact", ctypes.c_float), ("ms_total", ctypes.c_float)]

    def as_dict(self) -> dict:
        return {k: getattr(self, k) for k, _ in self._fields_}


class LaunchInfo(ctypes.Structure):
    _fields_ = [("phase", ctypes.c_uint32), ("path", ctypes.c_uint32), ("lanes", ctypes.c_uint32),
                ("rows", ctypes.c_uint32), ("param0", ctypes.c_uint32), ("param1", ctypes.c_uint32),
                ("ms", ctypes.c_float), ("lds_bytes", ctypes.c_uint32),
                ("nnz_a", ctypes.c_uint64), ("nprod", ctypes.c_uint64), ("nnz_c", ctypes.c_uint64)]

    def as_dict(self) -> dict:
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["phase"] = ("symbolic", "numeric")[d["phase"]]
        d["path"] = ("window", "hash", "global", "window_groups", "big_rows", "window_group_runs")[d["path"]]
        return d


# the same layout for numpy (a C struct of natural alignment: no padding between these fields)
_LAUNCH_DTYPE = np.dtype([("phase", "<u4"), ("path", "<u4"), ("lanes", "<u4"), ("rows", "<u4"), ("param0", "<u4"),
                          ("param1", "<u4"), ("ms", "<f4"), ("lds_bytes", "<u4"), ("nnz_a", "<u8"), ("nprod", "<u8"),
                          ("nnz_c", "<u8")])
assert _LAUNCH_DTYPE.itemsize == ctypes.sizeof(LaunchInfo)

_lib: Optional[ctypes.CDLL] = None


def load() -> ctypes.CDLL:
    """Load libspgemm_amd.so (built in-tree by spgemm_amd/csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C spgemm_amd/csrc` (hipcc, gfx950). "
            "spgemm_amd has no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    vp, u64, u32, i32 = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int
    pp = ctypes.POINTER(vp)
    sig = {
        "spgemm_version": (i32, []),
        "spgemm_last_error": (ctypes.c_char_p, []),
        "spgemm_device_count": (i32, [ctypes.POINTER(i32)]),
        "spgemm_create": (i32, [i32, pp]),
        "spgemm_destroy": (i32, [vp]),
        "spgemm_ctx_device": (i32, [vp, ctypes.POINTER(i32)]),
        "spgemm_set_stream": (i32, [vp, vp]),
        "spgemm_synchronize": (i32, [vp]),
        "spgemm_set_option": (i32, [vp, ctypes.c_char_p, ctypes.c_int64]),
        "spgemm_device_property": (i32, [vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64)]),
        "spgemm_mat_from_host": (i32, [vp, u64, u64, u64, vp, vp, vp, pp]),
        "spgemm_mat_wrap_device": (i32, [vp, u64, u64, u64, vp, vp, vp, i32, pp]),
        "spgemm_mat_invalidate": (i32, [vp, vp]),
        "spgemm_mat_info": (i32, [vp, ctypes.POINTER(u64), ctypes.POINTER(u64), ctypes.POINTER(u64)]),
        "spgemm_mat_device_ptrs": (i32, [vp, pp, pp, pp]),
        "spgemm_mat_to_host": (i32, [vp, vp, vp, vp, vp]),
        "spgemm_mat_rows_to_host": (i32, [vp, vp, u64, u64, vp, vp, vp, u64]),
        "spgemm_mat_check": (i32, [vp, vp, ctypes.POINTER(u64), ctypes.POINTER(u64)]),
        "spgemm_mat_free": (i32, [vp, vp]),
        "spgemm_multiply": (i32, [vp, vp, vp, pp]),
        "spgemm_symbolic": (i32, [vp, vp, vp, vp, ctypes.POINTER(u64)]),
        "spgemm_numeric": (i32, [vp, vp, vp, vp, vp, vp, ctypes.POINTER(u64)]),
        "spgemm_get_stats": (i32, [vp, ctypes.POINTER(Stats)]),
        "spgemm_get_launches": (i32, [vp, ctypes.POINTER(LaunchInfo), i32, ctypes.POINTER(i32)]),
        "spgemm_section_bitmap": (i32, [vp, vp, u32, vp]),
        "spgemm_partition_naive": (i32, [u32, u32, i32, ctypes.POINTER(Partition)]),
        "spgemm_partition_balanced": (i32, [vp, u32, u32, i32, ctypes.POINTER(Partition)]),
        "spgemm_row_nprod": (i32, [vp, vp, vp, vp]),
        "spgemm_shuffle_random": (i32, [u32, u64, vp]),
        "spgemm_shuffle_iterative": (i32, [vp, vp, u32, vp, vp, vp]),
        "spgemm_mat_relabel_columns": (i32, [vp, vp, vp, u64, pp]),
        "spgemm_mat_select": (i32, [vp, vp, vp, u64, vp, u64, i32, pp]),
        "spgemm_section_nnz": (i32, [vp, vp, u32, u64, u64, vp]),
        "spgemm_pack_sections": (i32, [vp, vp, u32, u64, u64, vp, vp, vp, vp, ctypes.POINTER(u64),
                                       ctypes.POINTER(u64)]),
        "spgemm_assemble_row_ptr": (i32, [vp, u64, u32, vp, vp, u64, vp, ctypes.POINTER(u64)]),
        "spgemm_buffer_alloc": (i32, [vp, ctypes.c_size_t, pp]),
        "spgemm_buffer_free": (i32, [vp, vp]),
        "spgemm_assemble_row_ptr_noalloc": (i32, [vp, u64, u32, vp, vp, u64, vp, ctypes.POINTER(u64)]),
        "spgemm_buffer_copy": (i32, [vp, vp, vp, vp, ctypes.c_size_t]),
        "spgemm_buffer_upload": (i32, [vp, vp, vp, ctypes.c_size_t]),
        "spgemm_upload_sections": (i32, [vp, u32, vp, vp, vp, vp, vp, vp, vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def _check(rc: int):
    if rc != 0:
        raise SpgemmError(rc, load().spgemm_last_error().decode("utf-8", "replace"))


def _np_ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def device_count() -> int:
    n = ctypes.c_int(0)
    rc = load().spgemm_device_count(ctypes.byref(n))
    return n.value if rc == 0 else 0


def bitmap_words(bits: np.ndarray) -> np.ndarray:
    """bool[n_sections] -> uint64 words (bit s = word s // 64, bit s % 64)."""
    bits = np.asarray(bits, dtype=bool)
    nwords = (len(bits) + 63) // 64
    padded = np.zeros(nwords * 64, dtype=np.uint8)
    padded[:len(bits)] = bits
    return np.packbits(padded, bitorder="little").view(np.uint64).copy()


def bitmap_bits(words: np.ndarray, n_sections: int) -> np.ndarray:
    return np.unpackbits(np.ascontiguousarray(words, dtype=np.uint64).view(np.uint8),
                         bitorder="little")[:n_sections].astype(bool)


class DeviceCsr:
    """Handle of a device-resident CSR (spgemm_mat)."""

    def __init__(self, ctx: "Context", handle: int, keepalive=None):
        self.ctx = ctx
        self.handle = ctypes.c_void_p(handle)
        self._keepalive = keepalive

    @property
    def shape(self) -> Tuple[int, int]:
        r, c, _ = self.info()
        return (r, c)

    def info(self) -> Tuple[int, int, int]:
        r, c, n = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        _check(load().spgemm_mat_info(self.handle, ctypes.byref(r), ctypes.byref(c), ctypes.byref(n)))
        return r.value, c.value, n.value

    @property
    def nnz(self) -> int:
        return self.info()[2]

    def device_ptrs(self) -> Tuple[int, int, int]:
        p, c, v = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        _check(load().spgemm_mat_device_ptrs(self.handle, ctypes.byref(p), ctypes.byref(c), ctypes.byref(v)))
        return p.value or 0, c.value or 0, v.value or 0

    def to_host(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        rows, _, nnz = self.info()
        row_ptr = np.empty(rows + 1, dtype=np.uint64)
        col_idx = np.empty(nnz, dtype=np.uint32)
        val = np.empty(nnz, dtype=np.float64)
        _check(load().spgemm_mat_to_host(self.ctx.handle, self.handle, _np_ptr(row_ptr), _np_ptr(col_idx),
                                         _np_ptr(val)))
        return row_ptr, col_idx, val

    def rows_to_host(self, row0: int, nrows: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """Rows [row0, row0 + nrows): (row_ptr rebased to 0, col_idx, val)."""
        # two calls: the first reads the row pointer to size the arrays
        full = np.empty(nrows + 1, dtype=np.uint64)
        lib = load()
        rc = lib.spgemm_mat_rows_to_host(self.ctx.handle, self.handle, row0, nrows, _np_ptr(full), None, None,
                                         ctypes.c_uint64(2 ** 63))
        _check(rc)
        n = int(full[-1] - full[0])
        col = np.empty(n, dtype=np.uint32)
        val = np.empty(n, dtype=np.float64)
        _check(lib.spgemm_mat_rows_to_host(self.ctx.handle, self.handle, row0, nrows, _np_ptr(full), _np_ptr(col),
                                           _np_ptr(val), n))
        return full - full[0], col, val

    def check(self) -> Tuple[int, int]:
        """(rows whose columns are not strictly increasing, stored zeros); raises when the row pointer
        or a column index is out of range."""
        a, b = ctypes.c_uint64(), ctypes.c_uint64()
        _check(load().spgemm_mat_check(self.ctx.handle, self.handle, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def invalidate(self):
        """Drop what the library cached about this matrix (call after rewriting wrapped arrays in place)."""
        _check(load().spgemm_mat_invalidate(self.ctx.handle, self.handle))

    def free(self):
        if self.handle:
            load().spgemm_mat_free(self.ctx.handle, self.handle)
            self.handle = ctypes.c_void_p(None)
            self._keepalive = None

    def __del__(self):
        try:
            if self.handle and self.ctx.handle:
                self.free()
        except Exception:
            pass


class Context:
    """One per process and GPU (spgemm_ctx)."""

    def __init__(self, device: int = 0):
        h = ctypes.c_void_p()
        _check(load().spgemm_create(device, ctypes.byref(h)))
        self.handle = h
        self.device = device

    def close(self):
        if self.handle:
            load().spgemm_destroy(self.handle)
            self.handle = ctypes.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name: str, value: int):
        _check(load().spgemm_set_option(self.handle, name.encode(), int(value)))

    def device_property(self, name: str) -> int:
        """What spgemm_create found out about the device ("lane_order_probe", "cu_count")."""
        v = ctypes.c_int64(0)
        _check(load().spgemm_device_property(self.handle, name.encode(), ctypes.byref(v)))
        return int(v.value)

    def set_stream(self, hip_stream: Optional[int]):
        """Run on the caller's hipStream_t, given as an integer handle (torch:
        `torch.cuda.current_stream().cuda_stream`).  Handle 0 is the device's default stream: it is
        passed on as hipStreamLegacy (1), because a NULL stream means "the context's own stream" to
        the library, and that one is created non-blocking and would not order with work queued on
        the default stream.  `None` restores the context's own stream."""
        if hip_stream is None:
            h = None
        else:
            h = int(hip_stream) or 1
        _check(load().spgemm_set_stream(self.handle, ctypes.c_void_p(h)))

    def synchronize(self):
        _check(load().spgemm_synchronize(self.handle))

    def stream_copy_rate(self, nbytes: int = 1 << 30, reps: int = 5) -> float:
        """GB/s (read + write) of a plain streaming copy on this GPU: the achievable HBM rate."""
        fn = load().spgemm_internal_stream_copy_rate
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
        out = ctypes.c_double(0.0)
        _check(fn(self.handle, nbytes, reps, ctypes.byref(out)))
        return out.value

    # -- matrices ---------------------------------------------------------------------------------
    def from_host(self, rows: int, cols: int, row_ptr, col_idx, val) -> DeviceCsr:
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint64)
        col_idx = np.ascontiguousarray(col_idx, dtype=np.uint32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        if len(row_ptr) != rows + 1:
            raise ValueError("row_ptr must have rows + 1 entries")
        nnz = len(col_idx)
        if len(val) != nnz:
            raise ValueError("col_idx and val differ in length")
        h = ctypes.c_void_p()
        _check(load().spgemm_mat_from_host(self.handle, rows, cols, nnz, _np_ptr(row_ptr), _np_ptr(col_idx),
                                           _np_ptr(val), ctypes.byref(h)))
        return DeviceCsr(self, h.value)

    def wrap_device(self, rows: int, cols: int, nnz: int, d_row_ptr: int, d_col_idx: int, d_val: int,
                    validate: bool = True, keepalive=None) -> DeviceCsr:
        h = ctypes.c_void_p()
        _check(load().spgemm_mat_wrap_device(self.handle, rows, cols, nnz, ctypes.c_void_p(d_row_ptr),
                                             ctypes.c_void_p(d_col_idx), ctypes.c_void_p(d_val), int(validate),
                                             ctypes.byref(h)))
        return DeviceCsr(self, h.value, keepalive)

    # -- multiply ---------------------------------------------------------------------------------
    def multiply(self, A: DeviceCsr, B: DeviceCsr) -> DeviceCsr:
        h = ctypes.c_void_p()
        _check(load().spgemm_multiply(self.handle, A.handle, B.handle, ctypes.byref(h)))
        return DeviceCsr(self, h.value)

    def symbolic(self, A: DeviceCsr, B: DeviceCsr, d_row_ptr: int) -> int:
        nnz = ctypes.c_uint64()
        _check(load().spgemm_symbolic(self.handle, A.handle, B.handle, ctypes.c_void_p(d_row_ptr), ctypes.byref(nnz)))
        return nnz.value

    def numeric(self, A: DeviceCsr, B: DeviceCsr, d_row_ptr: int, d_col_idx: int, d_val: int) -> int:
        nnz = ctypes.c_uint64()
        _check(load().spgemm_numeric(self.handle, A.handle, B.handle, ctypes.c_void_p(d_row_ptr),
                                     ctypes.c_void_p(d_col_idx), ctypes.c_void_p(d_val), ctypes.byref(nnz)))
        return nnz.value

    def stats(self) -> dict:
        s = Stats()
        _check(load().spgemm_get_stats(self.handle, ctypes.byref(s)))
        return s.as_dict()

    def launches(self) -> list:
        """Per-launch figures of the last multiply (needs set_option("profile_launches", 1))."""
        n = ctypes.c_int(0)
        _check(load().spgemm_get_launches(self.handle, None, 0, ctypes.byref(n)))
        arr = (LaunchInfo * max(n.value, 1))()
        _check(load().spgemm_get_launches(self.handle, arr, n.value, ctypes.byref(n)))
        return [arr[i].as_dict() for i in range(n.value)]

    def launch_ms(self) -> np.ndarray:
        """Only the durations of launches(), in the same order, without building a record per launch
        (for a timing loop that adds them up step by step)."""
        n = ctypes.c_int(0)
        _check(load().spgemm_get_launches(self.handle, None, 0, ctypes.byref(n)))
        arr = (LaunchInfo * max(n.value, 1))()
        _check(load().spgemm_get_launches(self.handle, arr, n.value, ctypes.byref(n)))
        rec = np.frombuffer(arr, dtype=_LAUNCH_DTYPE, count=n.value)
        return rec["ms"].astype(np.float64)

    # -- bitmap / partition / shuffle ------------------------------------------------------------
    def shuffle_iterative(self, C: DeviceCsr, iterations: int):
        """Iterative locality shuffle of the pattern of C (reference src/partition.cpp:123-202).
        Returns (row_shuffle, col_shuffle, row_variance)."""
        rows, cols, _ = C.info()
        rs = np.zeros(rows, dtype=np.uint32)
        cs = np.zeros(cols, dtype=np.uint32)
        var = np.zeros(max((iterations + 1) // 2, 1), dtype=np.float64)
        _check(load().spgemm_shuffle_iterative(self.handle, C.handle, int(iterations), _np_ptr(rs), _np_ptr(cs),
                                               _np_ptr(var)))
        return rs, cs, var[:(iterations + 1) // 2]

    def section_bitmap(self, A: DeviceCsr, n_sections: int) -> np.ndarray:
        words = np.zeros((n_sections + 63) // 64, dtype=np.uint64)
        _check(load().spgemm_section_bitmap(self.handle, A.handle, n_sections, _np_ptr(words)))
        return bitmap_bits(words, n_sections)

    def row_nprod(self, A: DeviceCsr, B: DeviceCsr) -> np.ndarray:
        out = np.zeros(A.info()[0], dtype=np.uint64)
        _check(load().spgemm_row_nprod(self.handle, A.handle, B.handle, _np_ptr(out)))
        return out

    def select(self, M: DeviceCsr, keep_rows=None, col_new_id=None, new_cols: int = 0,
               sort_rows: bool = True) -> DeviceCsr:
        kr = None if keep_rows is None else np.ascontiguousarray(keep_rows, dtype=np.uint32)
        cn = None if col_new_id is None else np.ascontiguousarray(col_new_id, dtype=np.uint32)
        if cn is not None and len(cn) != M.info()[1]:
            raise ValueError("col_new_id must have one entry per column of M")
        h = ctypes.c_void_p()
        _check(load().spgemm_mat_select(self.handle, M.handle, _np_ptr(kr), 0 if kr is None else len(kr),
                                        _np_ptr(cn), new_cols, int(sort_rows), ctypes.byref(h)))
        return DeviceCsr(self, h.value)

    def relabel_columns(self, M: DeviceCsr, d_new_id: int, new_cols: int) -> DeviceCsr:
        """M with column c renamed to new_id[c] (a device array given by its address, a bijection
        onto [0, new_cols)) and the rows sorted by the new ids."""
        h = ctypes.c_void_p()
        _check(load().spgemm_mat_relabel_columns(self.handle, M.handle, ctypes.c_void_p(d_new_id), int(new_cols),
                                                 ctypes.byref(h)))
        return DeviceCsr(self, h.value)

    # -- B exchange helpers -------------------------------------------------------------------------
    def section_nnz(self, M: DeviceCsr, n_sections: int, first_row: int, total_rows: int) -> np.ndarray:
        out = np.zeros(n_sections, dtype=np.uint64)
        _check(load().spgemm_section_nnz(self.handle, M.handle, n_sections, first_row, total_rows, _np_ptr(out)))
        return out

    def pack_sections(self, M: DeviceCsr, n_sections: int, first_row: int, total_rows: int, bits: np.ndarray,
                      d_row_len: int, d_col_idx: int, d_val: int) -> Tuple[int, int]:
        words = bitmap_words(bits)
        r, n = ctypes.c_uint64(), ctypes.c_uint64()
        _check(load().spgemm_pack_sections(self.handle, M.handle, n_sections, first_row, total_rows, _np_ptr(words),
                                           ctypes.c_void_p(d_row_len), ctypes.c_void_p(d_col_idx),
                                           ctypes.c_void_p(d_val), ctypes.byref(r), ctypes.byref(n)))
        return r.value, n.value

    def assemble_row_ptr(self, rows: int, n_sections: int, bits: np.ndarray, d_row_len_concat: int, n_row_len: int,
                         d_row_ptr_out: int, noalloc: bool = False, want_nnz: bool = True) -> Optional[int]:
        """noalloc=True: the variant that allocates nothing (at most 2^22 rows).  want_nnz=False: the
        call is only queued on the stream (nothing is read back) and returns None."""
        words = bitmap_words(bits)
        n = ctypes.c_uint64()
        fn = load().spgemm_assemble_row_ptr_noalloc if noalloc else load().spgemm_assemble_row_ptr
        _check(fn(self.handle, rows, n_sections, _np_ptr(words), ctypes.c_void_p(d_row_len_concat), n_row_len,
                  ctypes.c_void_p(d_row_ptr_out), ctypes.byref(n) if want_nnz else None))
        return n.value if want_nnz else None


# -- host-only entry points (no GPU needed) ------------------------------------------------------------
def partition_naive(height: int, width: int, n_parts: int):
    arr = (Partition * n_parts)()
    _check(load().spgemm_partition_naive(height, width, n_parts, arr))
    return [p.as_tuple() for p in arr]


def partition_balanced(row_weight: Sequence[int], width: int, n_parts: int):
    w = np.ascontiguousarray(row_weight, dtype=np.uint64)
    arr = (Partition * n_parts)()
    _check(load().spgemm_partition_balanced(_np_ptr(w), len(w), width, n_parts, arr))
    return [p.as_tuple() for p in arr]


def shuffle_random(n: int, seed: int) -> np.ndarray:
    out = np.zeros(n, dtype=np.uint32)
    _check(load().spgemm_shuffle_random(n, seed, _np_ptr(out)))
    return out
