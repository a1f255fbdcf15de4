"""GGML_OP_ARGMAX tie / NaN rule, pinned on the CPU: the restatement used by the GPU test (tests/test_ops_gpu.py::ref_argmax) against the reference
ggml CPU backend built from its own sources (oracle/_ref) -- the LAST of equal maxima wins, a NaN resets the running maximum
(R/ggml/src/ggml-cpu/ggml-cpu.c:2253-2261).  numpy.argmax (first of equals) is NOT the reference's rule."""
import numpy as np


def restated(x):
    out = []
    for row in np.asarray(x, np.float32):
        mx = np.float32(-np.inf); idx = 0
        for i, v in enumerate(row):
            mx = mx if mx > v else v
            if mx == v:
                idx = i
        out.append(idx)
    return np.asarray(out, np.int32)


def test_reference_argmax_takes_the_last_of_equal_maxima(ea, ref_cpu):
    rng = np.random.default_rng(7)
    x = rng.standard_normal((6, 257)).astype(np.float32)
    x[0, 3] = x[0, 200] = 9.0                      # tie
    x[1, :] = -np.inf                              # all equal to the initial maximum
    x[2, 10] = np.nan; x[2, 4] = 50.0              # the NaN forgets what came before it
    x[3, 256] = np.nan; x[3, 17] = 40.0            # a trailing NaN changes nothing
    x[4, :] = 1.5                                  # constant row
    g = ea.Graph(ref_cpu)
    a = g.tensor(ea.F32, x.shape[1], x.shape[0]); r = g.argmax(a)
    g.alloc(); g.set(a, x); g.compute()
    got = g.get(r, np.int32).reshape(-1)
    want = restated(x)
    assert np.array_equal(got, want)
    assert want[0] == 200 and want[1] == 256 and want[4] == 256 and want[3] == 17
    assert want[0] != int(np.argmax(x[0]))         # numpy's rule differs
