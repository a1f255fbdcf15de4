"""ctypes access to oracle/liboracle.so (the CPU restatement) -- TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from the product
package.  `load()` builds the library with gcc when it is missing or stale.
"""
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")
F32, F16, Q4_0, Q8_0, Q4_K, Q5_K, Q6_K = 0, 1, 2, 8, 12, 13, 14
BLOCK = {F32: (1, 4), F16: (1, 2), Q4_0: (32, 18), Q8_0: (32, 34), Q4_K: (256, 144), Q5_K: (256, 176), Q6_K: (256, 210)}
_lib = None


def build(force=False):
    srcs = [os.path.join(HERE, f) for f in ("ref_quants.c", "ref_ops.c", "oracle.h")]
    if force or not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", "-std=gnu11", "-Wall",
                               "-o", LIB, os.path.join(HERE, "ref_quants.c"), os.path.join(HERE, "ref_ops.c"), "-lm"])
    return LIB


def load():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        vp, i64, i32, f32 = C.c_void_p, C.c_int64, C.c_int, C.c_float
        L.orc_fp16_to_fp32.restype, L.orc_fp16_to_fp32.argtypes = f32, [C.c_uint16]
        L.orc_fp32_to_fp16.restype, L.orc_fp32_to_fp16.argtypes = C.c_uint16, [f32]
        for n in ("q8_0", "q8_1", "q8_K", "q4_0"):
            getattr(L, f"orc_quantize_row_{n}").argtypes = [vp, vp, i64]
        L.orc_dequantize_row.argtypes = [i32, vp, vp, i64]
        for n in ("q4_0_q8_0", "q8_0_q8_0", "q4_K_q8_K", "q5_K_q8_K", "q6_K_q8_K"):
            f = getattr(L, f"orc_vec_dot_{n}")
            f.restype, f.argtypes = f32, [i64, vp, vp]
        L.orc_mul_mat_q.argtypes = [i32, vp, vp, vp, i64, i64, i64]
        L.orc_mul_mat_f16.argtypes = [vp, i64, vp, vp, i64, i64, i64]
        L.orc_mul_mat_f32.argtypes = [vp, vp, vp, i64, i64, i64]
        L.orc_rms_norm.argtypes = [vp, vp, i64, i64, f32]
        L.orc_rope.argtypes = [vp, vp, vp, i64, i64, i64, i32, i32, f32, f32, f32, f32, f32, f32, i32]
        L.orc_soft_max.argtypes = [vp, vp, vp, i64, i64, i64, f32]
        L.orc_silu.argtypes = [vp, vp, i64]
        L.orc_relu.argtypes = [vp, vp, i64]
        L.orc_add.argtypes = [vp, vp, vp, i64, i64]
        L.orc_mul.argtypes = [vp, vp, vp, i64, i64]
        L.orc_cpy_f32_f16.argtypes = [vp, vp, i64]
        L.orc_attention.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i64, i64, i64, i64, i64, f32]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def row_bytes(t, k):
    b, s = BLOCK[t]
    return k // b * s


def quantize_q8_0(x):
    x = np.ascontiguousarray(x, np.float32); out = np.empty(x.size // 32 * 34, np.uint8)
    load().orc_quantize_row_q8_0(_p(x), _p(out), x.size); return out


def quantize_q8_1(x):
    x = np.ascontiguousarray(x, np.float32); out = np.empty(x.size // 32 * 36, np.uint8)
    load().orc_quantize_row_q8_1(_p(x), _p(out), x.size); return out


def quantize_q8_K(x):
    x = np.ascontiguousarray(x, np.float32); out = np.empty(x.size // 256 * 292, np.uint8)
    load().orc_quantize_row_q8_K(_p(x), _p(out), x.size); return out


def quantize_q4_0(x):
    x = np.ascontiguousarray(x, np.float32); out = np.empty(x.size // 32 * 18, np.uint8)
    load().orc_quantize_row_q4_0(_p(x), _p(out), x.size); return out


def dequantize(t, blocks, k):
    blocks = np.ascontiguousarray(blocks); out = np.empty(k, np.float32)
    load().orc_dequantize_row(t, _p(blocks), _p(out), k); return out


def mul_mat_q(t, w, x, k, rows):
    """w: raw block bytes [rows * row_bytes]; x: [T, k] f32 -> [T, rows] f32"""
    w = np.ascontiguousarray(w); x = np.ascontiguousarray(x, np.float32); T = x.size // k
    out = np.empty((T, rows), np.float32)
    load().orc_mul_mat_q(t, _p(w), _p(x), _p(out), k, rows, T); return out


def mul_mat_f16(a, x):
    """a: [rows, k] f16 (as uint16 or float16); x: [T, k] f32 -> [T, rows]"""
    a = np.ascontiguousarray(a).view(np.uint16); x = np.ascontiguousarray(x, np.float32)
    rows, k = a.shape; T = x.size // k; out = np.empty((T, rows), np.float32)
    load().orc_mul_mat_f16(_p(a), k, _p(x), _p(out), k, rows, T); return out


def mul_mat_f32(a, x):
    a = np.ascontiguousarray(a, np.float32); x = np.ascontiguousarray(x, np.float32)
    rows, k = a.shape; T = x.size // k; out = np.empty((T, rows), np.float32)
    load().orc_mul_mat_f32(_p(a), _p(x), _p(out), k, rows, T); return out


def rms_norm(x, eps):
    x = np.ascontiguousarray(x, np.float32); y = np.empty_like(x)
    load().orc_rms_norm(_p(x), _p(y), x.shape[-1], x.size // x.shape[-1], eps); return y


def rope(x, pos, n_dims, mode=0, freq_base=10000.0, freq_scale=1.0, ext_factor=0.0, attn_factor=1.0, beta_fast=32.0, beta_slow=1.0, n_ctx_orig=0):
    """x: [tokens, heads, ne0] f32"""
    x = np.ascontiguousarray(x, np.float32); pos = np.ascontiguousarray(pos, np.int32); y = np.empty_like(x)
    ne2, ne1, ne0 = x.shape
    load().orc_rope(_p(x), _p(pos), _p(y), ne0, ne1, ne2, n_dims, mode, freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow, n_ctx_orig)
    return y


def soft_max(x, mask, scale):
    """x: [ne02, ne01, nc]; mask: [>=ne01, nc] or None"""
    x = np.ascontiguousarray(x, np.float32); y = np.empty_like(x)
    ne02, ne01, nc = x.shape
    m = np.ascontiguousarray(mask, np.float32) if mask is not None else None
    load().orc_soft_max(_p(x), _p(m) if m is not None else None, _p(y), nc, ne01, ne02, scale); return y


def silu(x):
    x = np.ascontiguousarray(x, np.float32); y = np.empty_like(x); load().orc_silu(_p(x), _p(y), x.size); return y


def attention(q, k, v, mask, scale, H_kv):
    """q: [T, H, d] f32; k: [H_kv, n_kv, d] f16; v: [H_kv, d, n_kv] f16 (transposed cache); mask: [>=T, n_kv] f32 -> [T, H, d]"""
    q = np.ascontiguousarray(q, np.float32); k = np.ascontiguousarray(k).view(np.uint16); v = np.ascontiguousarray(v).view(np.uint16)
    mask = np.ascontiguousarray(mask, np.float32)
    T, H, d = q.shape; n_kv = k.shape[1]; out = np.empty_like(q)
    load().orc_attention(_p(q), _p(k), _p(v), _p(mask), _p(out), d, T, H, H_kv, n_kv, d, n_kv * d, n_kv, d * n_kv, mask.shape[1], scale)
    return out
