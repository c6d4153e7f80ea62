"""Numpy restatement of the device DEM sampler's random stream (slidingwindowdecoder_amd/csrc/swd_sampler.hip):
Philox4x32-10 keyed by the seed, counter (shot lo, shot hi, column // 4, 0), fault iff x < round(p 2^32).
Test infrastructure only."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(x, np.uint64) & np.uint64(0xFFFFFFFF) for x in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    for r in range(10):
        p0, p1 = M0 * c0, M1 * c2
        kk0, kk1 = np.uint64((k0 + r * W0) & 0xFFFFFFFF), np.uint64((k1 + r * W1) & 0xFFFFFFFF)
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ kk0
        n1 = p1 & np.uint64(0xFFFFFFFF)
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ kk1
        n3 = p0 & np.uint64(0xFFFFFFFF)
        c0, c1, c2, c3 = n0, n1, n2, n3
    return np.stack([c0, c1, c2, c3], axis=-1)


def sample_faults(priors, shots, seed, first_shot=0):
    n = len(priors)
    thr = np.minimum(np.floor(np.asarray(priors, np.float64) * 4294967296.0 + 0.5), 4294967295.0).astype(np.uint64)
    shot = (np.arange(shots, dtype=np.uint64) + np.uint64(first_shot))[:, None]
    grp = np.arange((n + 3) // 4, dtype=np.uint64)[None, :]
    x = philox4x32_10(shot & np.uint64(0xFFFFFFFF), shot >> np.uint64(32), grp, np.uint64(0), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    x = x.reshape(shots, -1)[:, :n]
    return (x < thr[None, :]).astype(np.uint8)
