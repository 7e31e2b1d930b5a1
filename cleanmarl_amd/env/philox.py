"""numpy Philox4x32-10 -- bit-identical to cm_philox4x32 in csrc/cm_common.h."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
STREAM_ACT, STREAM_ENV_RESET, STREAM_ENV_STEP = 1, 2, 3


def philox4x32(c0, c1, c2, c3, k0, k1):
    """Vectorised over numpy arrays of uint32 counters; returns 4 uint32 arrays."""
    with np.errstate(over="ignore"):
        c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in np.broadcast_arrays(c0, c1, c2, c3))
        k0 = np.uint32(k0)
        k1 = np.uint32(k1)
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ k0
            n1 = p1.astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ k1
            n3 = p0.astype(np.uint32)
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def u01(x):
    """uniform in [0,1) from the top 24 bits (exact in fp32)."""
    return (np.asarray(x, dtype=np.uint32) >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def split_seed(seed):
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return seed & 0xFFFFFFFF, seed >> 32
