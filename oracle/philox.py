"""Philox4x32-10 counter RNG -- TEST INFRASTRUCTURE (checker only; nothing under cleanmarl_amd/ imports this).

The reference draws its actions from torch's global generator (``Categorical(...).sample()``, cleanmarl/mappo_multienvs.py:172-176)
and re-seeds every env worker from OS entropy (``random.randint`` in a forked child, :251): its streams cannot be reproduced.  The build
keys every draw by (seed, global row, time step, stream id) with Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers:
as easy as 1, 2, 3", SC'11; reference implementation Random123 1.x, philox.h).  This file restates the PUBLISHED algorithm with plain
Python integers, independently of the product's two implementations (csrc/cm_common.h::cm_philox4x32, cleanmarl_amd/env/philox.py):

    round:   (hi0, lo0) = M0 * c0,  (hi1, lo1) = M1 * c2   (32 x 32 -> 64-bit products)
             c <- (hi1 ^ c1 ^ k0,  lo1,  hi0 ^ c3 ^ k1,  lo0)
    key:     k <- (k0 + W0, k1 + W1)  between rounds (bumped 9 times for 10 rounds)
    M0 = 0xD2511F53, M1 = 0xCD9E8D57, W0 = 0x9E3779B9 (golden ratio), W1 = 0xBB67AE85 (sqrt 3 - 1)

PIN: ``KAT`` below holds the three philox4x32_10 known-answer vectors of Random123's ``kat_vectors`` file (counter, key -> output);
tests/test_pins.py holds this restatement, the numpy twin, the library's host function and the device kernel to them.
"""
import numpy as np

M0, M1 = 0xD2511F53, 0xCD9E8D57
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF
STREAM_ACT = 1  # 4th counter word of the action sampler's draws (csrc/cm_common.h: CM_STREAM_ACT)

# Random123 kat_vectors, lines "philox4x32 10 <c0 c1 c2 c3> <k0 k1>   <expected r0 r1 r2 r3>"
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def philox4x32_10(ctr, key):
    """ctr = (c0, c1, c2, c3), key = (k0, k1), python ints in [0, 2^32) -> (r0, r1, r2, r3)."""
    c0, c1, c2, c3 = (int(c) & MASK for c in ctr)
    k0, k1 = (int(k) & MASK for k in key)
    for rnd in range(10):
        if rnd:
            k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
        p0, p1 = M0 * c0, M1 * c2
        c0, c1, c2, c3 = (p1 >> 32) ^ c1 ^ k0, p1 & MASK, (p0 >> 32) ^ c3 ^ k1, p0 & MASK
    return c0, c1, c2, c3


def u01(x):
    """The build's uniform in [0, 1): the top 24 bits of a word (exactly representable in fp32)."""
    return np.float32((int(x) >> 8) * (1.0 / 16777216.0))


def act_uniforms(n_rows, seed, row_offset, t):
    """The action sampler's uniform of every row: first output word of the block keyed by the 64-bit seed, counted by
    (row low word, row high word, t, STREAM_ACT) with row = row_offset + local row.  -> float32 [n_rows]"""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    key = (seed & MASK, seed >> 32)
    out = np.empty(n_rows, np.float32)
    for r in range(n_rows):
        row = (int(row_offset) + r) & 0xFFFFFFFFFFFFFFFF
        out[r] = u01(philox4x32_10((row & MASK, row >> 32, int(t) & MASK, STREAM_ACT), key)[0])
    return out
