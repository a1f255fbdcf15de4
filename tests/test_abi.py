"""The plugin is compiled against include/ggml_abi.h only; these tests keep that header honest.  CPU only."""
import json
import os
import subprocess
import ctypes as C
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_GGML = "/root/reference/llama.cpp/ggml"
PROBE = os.path.join(ROOT, "tests", "abi_probe.cpp")
GOLD = os.path.join(ROOT, "tests", "golden", "abi_layout.json")


def _probe(tmp_path, flags, name):
    exe = str(tmp_path / name)
    subprocess.check_call(["g++", "-std=c++17", "-w", "-O1"] + flags + [PROBE, "-o", exe])
    return json.loads(subprocess.check_output([exe]))


def test_our_abi_header_matches_recorded_reference_layout(tmp_path):
    ours = _probe(tmp_path, [f"-I{ROOT}/include"], "probe_ours")
    assert ours == json.load(open(GOLD))


@pytest.mark.skipif(not os.path.isdir(REF_GGML), reason="reference tree not present")
def test_recorded_layout_matches_live_reference_headers(tmp_path):
    ref = _probe(tmp_path, ["-DUSE_REFERENCE", f"-I{REF_GGML}/include", f"-I{REF_GGML}/src",
                            f"-L{ROOT}/oracle/_ref", "-lggml-ref", f"-Wl,-rpath,{ROOT}/oracle/_ref"], "probe_ref")
    assert ref == json.load(open(GOLD))


def test_plugin_loads_and_exports_every_declared_symbol(ea):
    """C-ABI library loads without a GPU and exports what include/ggml_mi355x.h declares (no compute calls)."""
    lib = C.CDLL(ea.require_plugin())
    import re
    hdr = open(os.path.join(ROOT, "include", "ggml_mi355x.h")).read()
    declared = re.findall(r"GGML_MI355X_API\s+[\w\s\*]+?\b(ggml_backend_\w+)\s*\(", hdr)
    assert set(declared) >= {"ggml_backend_init", "ggml_backend_score", "ggml_backend_mi355x_reg",
                             "ggml_backend_mi355x_device_count", "ggml_backend_mi355x_split_buffer_type"}
    for sym in declared:
        assert hasattr(lib, sym), sym


def test_host_library_loads(ea):
    h = ea.host()
    for sym in ("eh_backend_load", "eh_ctx_new", "eh_mul_mat", "eh_compute"):
        assert hasattr(h, sym)


def test_registry_without_gpu_reports_zero_score_or_devices(ea):
    """ggml_backend_score() must be 0 where no gfx950 device exists so a reference loader skips us
    (R/ggml/src/ggml-backend-reg.cpp:229-235); with a GPU it must be > 0."""
    lib = C.CDLL(ea.require_plugin())
    lib.ggml_backend_score.restype = C.c_int
    lib.ggml_backend_mi355x_device_count.restype = C.c_int
    score, n = lib.ggml_backend_score(), lib.ggml_backend_mi355x_device_count()
    assert (score > 0) == (n > 0)
