"""CPU (reference ggml backend) check of the fused draft chain: the host logic of Model::decode_chain -- KV slots of all steps found
up front, masks, the ARGMAX -> GET_ROWS hand-off -- gives the same drafts as the stepwise loop.  No GPU involved."""
import os
import numpy as np
import pytest


def test_fused_chain_equals_stepwise_on_reference_cpu(ea, ref_cpu, monkeypatch):
    res = []
    for stepwise in (False, True):
        if stepwise: monkeypatch.setenv("EH_STEPWISE_DRAFT", "1")
        else: monkeypatch.delenv("EH_STEPWISE_DRAFT", raising=False)
        t = ea.Model(ref_cpu, "tiny", "q4_k_m", n_ctx=256, seed=5)
        d = ea.Model(ref_cpu, "tiny", "q4_k_m", n_ctx=256, eagle_of=t, seed=5, accept_p=0.75)
        prompt = [int(x) for x in np.random.default_rng(3).integers(5, 512, 16)]
        spec, st = ea.spec_generate(t, d, prompt, 40, n_draft=4)
        res.append((spec, st["n_accept"], st["n_drafted"], st["n_iters"], st["n_draft_calls"]))
        d.close(); t.close()
    assert res[0][:4] == res[1][:4]
    assert res[0][4] < res[1][4]


def test_decode_rejects_out_of_vocabulary_tokens(ea, ref_cpu):
    """llama_decode fails on an invalid token id (R/src/llama.cpp:9500); the host mirror must not read past token_embd"""
    t = ea.Model(ref_cpu, "tiny", "q4_k_m", n_ctx=64, seed=5)
    with pytest.raises(RuntimeError):
        t.decode([5, 512], [0, 1])
    with pytest.raises(RuntimeError):
        t.decode([-1], [0])
    lg, _ = t.decode([5, 511], [0, 1])
    assert np.isfinite(lg).all()
    t.close()


def test_speculation_is_lossless_on_reference_cpu(ea, ref_cpu):
    """greedy speculative decoding (fused draft chain, device-side arg-max ops on the CPU backend too) reproduces plain greedy decoding"""
    t = ea.Model(ref_cpu, "tiny", "q4_0", n_ctx=256, seed=9)
    d = ea.Model(ref_cpu, "tiny", "q4_0", n_ctx=256, eagle_of=t, seed=9, accept_p=0.6)
    prompt = [int(x) for x in np.random.default_rng(4).integers(5, 512, 12)]
    plain, _ = ea.plain_generate(t, prompt, 32)
    spec, st = ea.spec_generate(t, d, prompt, 32, n_draft=3)
    assert plain == spec[:len(plain)] and st["n_accept"] > 0
    d.close(); t.close()
