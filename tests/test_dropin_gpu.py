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


@pytest.mark.parametrize("fusion", ["fused", "unfused", "flash_attn", "split_row", "split_row_staged", "cut_qcur", "cut_norms"])
def test_reference_llama_decode_on_plugin(ea, tmp_path, fusion):
    env = dict(os.environ)
    if fusion == "unfused":
        env["GGML_MI355X_NO_FUSION"] = "1"
    if fusion == "flash_attn":                       # the reference's example commands use -fa: FLASH_ATTN_EXT runs on the plugin too
        env["DROPIN_FLASH_ATTN"] = "1"
    if fusion.startswith("split_row"):
        # -sm row under the reference's own loader (llama-model.cpp:304-326 -> get_proc_address("ggml_backend_split_buffer_type")):
        # every mat-mul weight sliced over three logical devices of the one GPU; "staged" takes the no-peer-access return path
        env["DROPIN_SPLIT_ROW"] = "1"
        env["GGML_MI355X_SPLIT_FAKE_DEVICES"] = "3"
        if fusion == "split_row_staged":
            env["GGML_MI355X_SPLIT_STAGE"] = "1"
    if fusion == "cut_qcur":
        # the scheduler's eval callback asks for Qcur-* and ffn_gate-*: graph views end between wq and wk / between gate and up, the
        # host reads the tensors there; a norm folded into the first launch has to be in memory for the next view (ggml-backend.cpp:1402-1430)
        env["DROPIN_CUT"] = "Qcur,ffn_gate"
    if fusion == "cut_norms":
        env["DROPIN_CUT"] = "attn_norm,ffn_norm,ffn_gate"
    out = subprocess.run([BIN, ea.require_plugin(), str(tmp_path)], env=env, capture_output=True, text=True, timeout=600)
    txt = out.stdout + out.stderr
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", f"dropin_{fusion}.log"), "w").write(txt)
    lines = [l for l in out.stdout.splitlines() if "rel-L2" in l or "DROP-IN" in l or "loaded backend" in l]
    if fusion.startswith("split_row"):               # the weights really went through the split buffer type
        assert "MI355X_Split" in txt, txt[-3000:]
    print("\n".join(lines))
    assert out.returncode == 0, txt[-4000:]
    assert "DROP-IN OK" in out.stdout
