"""ctypes access to functions of the REAL reference build (oracle/_ref/libggml-ref*.so) -- tests only."""
import ctypes as C
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
F32, F16, Q4_0, Q8_0, Q4_K, Q5_K, Q6_K = 0, 1, 2, 8, 12, 13, 14
BLOCK = {Q4_0: (32, 18), Q8_0: (32, 34), Q4_K: (256, 144), Q5_K: (256, 176), Q6_K: (256, 210)}
_libs = {}
REF_GGML_PATH = os.path.join(REF, "libggml-ref.so")
REF_GGML_SCALAR_PATH = os.path.join(REF, "libggml-ref-scalar.so")


def reference_cpu(ea, threads=None, scalar=False):
    """The reference's own ggml CPU backend (oracle/_ref, built from /root/reference by oracle/Makefile) loaded through the same host ABI
    as the plugin.  Checker / CPU baseline only: tests, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
    p = REF_GGML_SCALAR_PATH if scalar else REF_GGML_PATH
    if not os.path.exists(p):
        raise FileNotFoundError(f"{p} not built (make -C oracle ref; needs /root/reference)")
    if threads is None:      # cores this process may actually use (a GPU box exposes many more than its share), capped
        try:
            threads = len(os.sched_getaffinity(0))
        except AttributeError:
            threads = os.cpu_count() or 4
        threads = max(1, min(threads, 16))
    return ea.Backend(p, "ggml_backend_cpu_reg", 0, threads)



def lib(scalar=True):
    key = "scalar" if scalar else "avx2"
    if key not in _libs:
        p = os.path.join(REF, "libggml-ref-scalar.so" if scalar else "libggml-ref.so")
        L = C.CDLL(p)
        vp, i64 = C.c_void_p, C.c_int64
        L.ggml_quantize_chunk.restype = C.c_size_t
        L.ggml_quantize_chunk.argtypes = [C.c_int, vp, vp, i64, i64, i64, vp]
        L.ggml_fp16_to_fp32.restype, L.ggml_fp16_to_fp32.argtypes = C.c_float, [C.c_uint16]
        L.ggml_fp32_to_fp16.restype, L.ggml_fp32_to_fp16.argtypes = C.c_uint16, [C.c_float]
        for n in ("quantize_row_q8_0_ref", "quantize_row_q8_1_ref", "quantize_row_q8_K_ref", "quantize_row_q4_0_ref",
                  "dequantize_row_q4_0", "dequantize_row_q8_0", "dequantize_row_q4_K", "dequantize_row_q5_K", "dequantize_row_q6_K"):
            getattr(L, n).argtypes = [vp, vp, i64]
        for n in ("q4_0_q8_0", "q8_0_q8_0", "q4_K_q8_K", "q5_K_q8_K", "q6_K_q8_K"):
            getattr(L, f"ggml_vec_dot_{n}").argtypes = [C.c_int, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t, C.c_int]
        L.ggml_cpu_init.argtypes = []
        L.ggml_cpu_init()
        _libs[key] = L
    return _libs[key]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def quantize(t, x, n_per_row, scalar=True):
    """Reference weight quantiser (ggml_quantize_chunk) -> raw block bytes."""
    x = np.ascontiguousarray(x, np.float32)
    rows = x.size // n_per_row
    b, s = BLOCK[t]
    out = np.empty(rows * (n_per_row // b) * s, np.uint8)
    n = lib(scalar).ggml_quantize_chunk(t, _p(x), _p(out), 0, rows, n_per_row, None)
    assert n == out.size
    return out


def quantize_act(name, x, scalar=True):
    x = np.ascontiguousarray(x, np.float32)
    per = {"q8_0": (32, 34), "q8_1": (32, 36), "q8_K": (256, 292)}[name]
    out = np.empty(x.size // per[0] * per[1], np.uint8)
    getattr(lib(scalar), f"quantize_row_{name}_ref")(_p(x), _p(out), x.size)
    return out


def dequantize(t, blocks, k, scalar=True):
    name = {Q4_0: "q4_0", Q8_0: "q8_0", Q4_K: "q4_K", Q5_K: "q5_K", Q6_K: "q6_K"}[t]
    blocks = np.ascontiguousarray(blocks)
    out = np.empty(k, np.float32)
    getattr(lib(scalar), f"dequantize_row_{name}")(_p(blocks), _p(out), k)
    return out


def vec_dot(name, n, x, y, scalar=True):
    s = C.c_float(0)
    getattr(lib(scalar), f"ggml_vec_dot_{name}")(n, C.byref(s), 0, _p(np.ascontiguousarray(x)), 0, _p(np.ascontiguousarray(y)), 0, 1)
    return np.float32(s.value)
