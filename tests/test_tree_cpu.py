"""Tree speculative driver (host/tree_driver.cpp, mirror of R/examples/speculative/speculative-eagle.cpp:232-670) on the reference
CPU backend: branch forks really happen, the KV fix-up keeps both caches consistent, and greedy verification stays lossless.  No GPU."""
import numpy as np
import pytest


def _models(ea, be, ftype="q4_k_m", accept_p=0.7, seed=5, predictable=True, n_ctx=512):
    t = ea.Model(be, "tiny", ftype, n_ctx=n_ctx, seed=seed, predictable=predictable)
    d = ea.Model(be, "tiny", ftype, n_ctx=n_ctx, eagle_of=t, seed=seed, accept_p=accept_p, predictable=predictable)
    return t, d


def test_tree_driver_forks_and_stays_lossless(ea, ref_cpu):
    """temp = 0 (greedy verification), temp_dft = 1.5 (flat draft distribution => candidates beyond the first exceed p_split): the emitted
    tokens are exactly plain greedy decoding's, forks happen, and more than one token per round is accepted on average"""
    t, d = _models(ea, ref_cpu)
    prompt = [int(x) for x in np.random.default_rng(3).integers(5, 512, 16)]
    plain, _ = ea.plain_generate(t, prompt, 48)
    s = ea.TreeSession(t, d, prompt, n_seq_dft=4, n_draft=6, p_split=0.02, temp=0.0, temp_dft=1.5, top_k=8)
    toks, st = s.run(48)
    s.close()
    assert toks[:48] == plain[:48]
    assert st["n_forks"] > 0 and st["max_batch"] > 2
    assert st["n_accept"] > 0 and st["n_predict"] == len(toks)
    d.close(); t.close()


def test_tree_driver_chain_limit_equals_chain_driver(ea, ref_cpu):
    """temp_dft = 0: one-hot candidates, p_split never exceeded -- the tree degenerates to the chain, as the reference does at --temp 0"""
    t, d = _models(ea, ref_cpu, accept_p=0.8)
    prompt = [int(x) for x in np.random.default_rng(7).integers(5, 512, 12)]
    plain, _ = ea.plain_generate(t, prompt, 40)
    s = ea.TreeSession(t, d, prompt, n_seq_dft=4, n_draft=5, p_split=0.1, temp=0.0, temp_dft=0.0)
    toks, st = s.run(40)
    s.close()
    assert toks[:40] == plain[:40]
    assert st["n_forks"] == 0 and st["n_accept"] > 0
    d.close(); t.close()


def test_tree_driver_stochastic_is_reproducible(ea, ref_cpu):
    """temp > 0: stochastic acceptance (r <= p_tgt / p_dft, residual resampling) with a seeded generator: same seed, same tokens"""
    outs = []
    for _ in range(2):
        t, d = _models(ea, ref_cpu, predictable=False, seed=11)
        prompt = [int(x) for x in np.random.default_rng(9).integers(5, 512, 10)]
        s = ea.TreeSession(t, d, prompt, n_seq_dft=3, n_draft=4, p_split=0.05, temp=0.8, top_k=16, seed=77)
        toks, st = s.run(24)
        s.close(); d.close(); t.close()
        outs.append((toks, st["n_accept"], st["n_forks"]))
        assert len(toks) >= 24 and all(0 <= x < 512 for x in toks)
    assert outs[0] == outs[1]
