"""Row-split buffer type (-sm row, csrc/split.cpp; reference get_row_split, R/ggml/src/ggml-cuda/ggml-cuda.cu:735-748): the slicing
arithmetic through the exported C symbol.  CPU only -- no device is touched."""
import ctypes as C
import numpy as np


def _split(ea, nrows, props, n_dev):
    lib = C.CDLL(ea.require_plugin())
    f = lib.ggml_backend_mi355x_row_split
    f.argtypes = [C.c_int64, C.POINTER(C.c_float), C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    arr = (C.c_float * 16)(*(list(props) + [0.0] * (16 - len(props)))) if props is not None else None
    out = []
    for d in range(n_dev):
        lo, hi = C.c_int64(), C.c_int64()
        f(nrows, arr, n_dev, d, C.byref(lo), C.byref(hi))
        out.append((lo.value, hi.value))
    return out


def test_even_split_covers_all_rows_in_whole_tiles(ea):
    for nrows, n_dev in [(4096, 8), (11008, 8), (32000, 4), (28672, 8), (13824, 2), (4096, 1), (100, 2)]:
        r = _split(ea, nrows, None, n_dev)
        assert r[0][0] == 0 and r[-1][1] == nrows
        for (lo, hi), (lo2, _) in zip(r, r[1:]):
            assert hi == lo2 and lo <= hi and hi % 128 == 0      # contiguous, boundaries on the rounding (16-row tiles stay whole)
        assert sum(hi - lo for lo, hi in r) == nrows


def test_proportional_split_follows_tensor_split(ea):
    r = _split(ea, 4096, [3, 1, 0, 4], 4)                        # -ts 3,1,0,4
    assert r == [(0, 1536), (1536, 2048), (2048, 2048), (2048, 4096)]
    r = _split(ea, 11008, [1, 1], 2)
    assert r == [(0, 5504 - 5504 % 128), (5504 - 5504 % 128, 11008)]
    assert _split(ea, 4096, [0, 0, 0, 0], 4) == _split(ea, 4096, None, 4)     # all zero = even (reference: default split)
