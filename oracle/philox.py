"""CPU restatement of the counter-based normal generator of csrc/misc.hip (oracle; test infrastructure).

Not part of the reference (which draws eps with torch's global generator, DeepGPLayer.__call__
`Normal(...).rsample()`); it exists so that data-parallel ranks draw the same eps a single GPU would
(SURVEY 8e).  Philox4x32-10 (Salmon et al., SC'11) + Box-Muller, keyed exactly as the kernel:
  counter = (row_lo, row_hi, s | (c // 4) << 20, stream_lo), key = (seed_lo, seed_hi ^ stream_hi)
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c, k0, k1):
    """c: (4, N) uint32 counters; k0, k1: uint32 scalars.  Returns (4, N) uint32."""
    c = [x.astype(np.uint32) for x in c]
    k0, k1 = np.uint32(k0), np.uint32(k1)
    with np.errstate(over='ignore'):
        for _ in range(10):
            p0 = M0 * c[0].astype(np.uint64)
            p1 = M1 * c[2].astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & MASK).astype(np.uint32)
            c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
            k0, k1 = np.uint32(k0 + W0), np.uint32(k1 + W1)
    return c


def normal(seed, stream_id, row0, S, n, b):
    """eps:(S, n, b) float64, identical (to rounding) to nsgp_philox_normal_f64."""
    bq = (b + 3) // 4
    s, i, cq = np.meshgrid(np.arange(S), np.arange(n), np.arange(bq), indexing='ij')
    row = (row0 + i).astype(np.uint64).ravel()
    c = [(row & MASK).astype(np.uint32), (row >> np.uint64(32)).astype(np.uint32),
         (s.ravel().astype(np.uint32) | (cq.ravel().astype(np.uint32) << np.uint32(20))),
         np.full(row.shape, np.uint32(stream_id & 0xFFFFFFFF), dtype=np.uint32)]
    out = philox4x32_10(c, seed & 0xFFFFFFFF, ((seed >> 32) ^ (stream_id >> 32)) & 0xFFFFFFFF)
    z = np.empty((4, row.size))
    for q in range(2):
        u1 = (out[2 * q].astype(np.float64) + 0.5) * 2.0 ** -32
        u2 = (out[2 * q + 1].astype(np.float64) + 0.5) * 2.0 ** -32
        r = np.sqrt(-2.0 * np.log(u1))
        th = 2.0 * np.pi * u2
        z[2 * q], z[2 * q + 1] = r * np.cos(th), r * np.sin(th)
    z = z.T.reshape(S, n, bq * 4)
    return z[:, :, :b]
