"""Synthetic quantised weights for tests: random but VALID blocks of each format (every bit pattern of the
quant fields is legal; the fp16 scales are drawn so that dequantised weights are O(0.02)), no quantiser needed."""
import numpy as np

Q4_0, Q8_0, Q4_K, Q5_K, Q6_K = 2, 8, 12, 13, 14
BLOCK = {Q4_0: (32, 18), Q8_0: (32, 34), Q4_K: (256, 144), Q5_K: (256, 176), Q6_K: (256, 210)}


def random_blocks(t, rows, k, rng, scale=0.02):
    b, s = BLOCK[t]
    nb = rows * (k // b)
    raw = rng.integers(0, 256, (nb, s), dtype=np.uint8)
    def h(v):
        return np.asarray(v, np.float16).view(np.uint8).reshape(nb, 2)
    u = rng.uniform(0.5, 1.5, nb)
    if t == Q4_0:
        raw[:, 0:2] = h(scale / 4 * u)
    elif t == Q8_0:
        raw[:, 0:2] = h(scale / 64 * u)
    elif t in (Q4_K, Q5_K):
        q = 8.0 if t == Q4_K else 16.0
        raw[:, 0:2] = h(scale / (32 * q) * u)
        raw[:, 2:4] = h(scale / 32 * rng.uniform(0.5, 1.5, nb))
    elif t == Q6_K:
        raw[:, 208:210] = h(scale / (64 * 16) * u)
    return raw.reshape(-1)
