"""The weight re-layout (csrc/tile_layout.h) is a pure byte permutation inside a 16-row group: checked here on
the CPU by compiling the very header the device kernels use.  CPU only."""
import ctypes as C
import os
import subprocess
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TYPES = {"q4_0": 2, "q8_0": 8, "q4_K": 12, "q5_K": 13, "q6_K": 14}
UNIT = {"q4_0": 144, "q8_0": 272, "q4_K": 144, "q5_K": 176, "q6_K": 210}


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("tile") / "tile_probe.so")
    subprocess.check_call(["g++", "-x", "c++", "-O1", "-shared", "-fPIC", f"-I{ROOT}/eagle-in-llama.cpp_amd/csrc",
                           os.path.join(ROOT, "tests", "tile_probe.c"), "-o", so])
    lib = C.CDLL(so)
    for f in ("probe_unit_bytes", "probe_tile_bytes", "_Z14probe_tile_srciiPiS_"):
        pass
    return lib


def _sym(lib, name):
    # compiled as C++: resolve the mangled names once
    out = subprocess.check_output(["nm", "-D", lib._name]).decode().split()
    for s in out:
        if name in s:
            return getattr(lib, s)
    raise KeyError(name)


def tile_map(lib, t):
    f = _sym(lib, "probe_tile_src")
    f.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    tb = _sym(lib, "probe_tile_bytes"); tb.restype = C.c_int
    nbytes = tb(t)
    n, sb = C.c_int(), C.c_int()
    m = np.zeros((nbytes // 2, 2), np.int32)
    for h in range(nbytes // 2):
        f(t, 2 * h, C.byref(n), C.byref(sb))
        m[h] = (n.value, sb.value)
    return m


@pytest.mark.parametrize("tname", list(TYPES))
def test_tile_is_a_permutation_of_16_rows_by_one_unit(probe, tname):
    t, ub = TYPES[tname], UNIT[tname]
    ub_f = _sym(probe, "probe_unit_bytes"); ub_f.restype = C.c_int
    assert ub_f(t) == ub
    m = tile_map(probe, t)
    assert len(m) == 16 * ub // 2
    assert (m[:, 0] >= 0).all() and (m[:, 0] < 16).all() and (m[:, 1] >= 0).all() and (m[:, 1] < ub).all() and (m[:, 1] % 2 == 0).all()
    flat = m[:, 0] * ub + m[:, 1]
    assert len(np.unique(flat)) == len(flat)              # every source half-word exactly once


def test_q4_K_lane_fragments(probe):
    """lane (n, kq) of the kernel reads hdr at 16n and qs piece ga at 256 + 1024 ga + 16 lane: bytes
    qs[64 (kq>>1) + 16 (kq&1) + 32 ga ..+16) of row n (block_q4_K: d, dmin, scales[12], qs[128] -- R/ggml/src/ggml-common.h)"""
    m = tile_map(probe, 12)
    for lane in range(64):
        n, kq = lane & 15, lane >> 4
        for ga in range(2):
            h0 = (256 + 1024 * ga + 16 * lane) // 2
            want = 16 + 64 * (kq >> 1) + 16 * (kq & 1) + 32 * ga
            assert (m[h0:h0 + 8, 0] == n).all() and (m[h0:h0 + 8, 1] == want + 2 * np.arange(8)).all()
    for n in range(16):
        assert (m[8 * n:8 * n + 8, 0] == n).all() and (m[8 * n:8 * n + 8, 1] == 2 * np.arange(8)).all()


def test_q6_K_and_q8_0_scales_are_grouped_per_row(probe):
    m = tile_map(probe, 14)
    for n in range(16):
        assert tuple(m[(3328 + 2 * n) // 2]) == (n, 208)                    # f16 d of row n
        h0 = (3072 + 16 * n) // 2
        assert (m[h0:h0 + 8, 0] == n).all() and (m[h0:h0 + 8, 1] == 192 + 2 * np.arange(8)).all()
    m = tile_map(probe, 8)
    for n in range(16):
        h0 = (4096 + 16 * n) // 2
        assert (m[h0:h0 + 8, 0] == n).all() and (m[h0:h0 + 8, 1] == 34 * np.arange(8)).all()      # the eight block scales of row n
