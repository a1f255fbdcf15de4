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
