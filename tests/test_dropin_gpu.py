"""Drop-in proof under the reference's own stack: oracle/_ref/dropin_llama (tests/dropin_llama.cpp compiled against the
reference's llama.h and linked with the reference's libllama + ggml, both built unmodified from /root/reference) loads
the plugin with ggml_backend_load() and runs llama_decode / llama_decode_initial / llama_decode_draft with every layer
on the MI355X device vs. on the reference CPU backend.  The reference's scheduler and graph allocator (buffer
re-use) are live here, which is what validates the executor's dependency-checked fusions."""
import os
import subprocess
import pytest

from conftest import ROOT

BIN = os.path.join(ROOT, "oracle", "_ref", "dropin_llama")
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(BIN), reason="oracle/_ref/dropin_llama not built")]


@pytest.mark.parametrize("fusion", ["fused", "unfused", "flash_attn"])
def test_reference_llama_decode_on_plugin(ea, tmp_path, fusion):
    env = dict(os.environ)
    if fusion == "unfused":
        env["GGML_MI355X_NO_FUSION"] = "1"
    if fusion == "flash_attn":                       # the reference's example commands use -fa: FLASH_ATTN_EXT runs on the plugin too
        env["DROPIN_FLASH_ATTN"] = "1"
    out = subprocess.run([BIN, ea.require_plugin(), str(tmp_path)], env=env, capture_output=True, text=True, timeout=600)
    txt = out.stdout + out.stderr
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", f"dropin_{fusion}.log"), "w").write(txt)
    lines = [l for l in out.stdout.splitlines() if "rel-L2" in l or "DROP-IN" in l or "loaded backend" in l]
    print("\n".join(lines))
    assert out.returncode == 0, txt[-4000:]
    assert "DROP-IN OK" in out.stdout
