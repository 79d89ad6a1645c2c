"""Second, independent restatement of the reference op semantics in numpy /
pure Python, used only to cross-check the C oracle on small cases.

It is deliberately formulated differently from oracle/eda_oracle.c:
FPS picks the winner with the closed-form tie rule of SURVEY.md Appendix A2
(min over (bitreverse(k mod bs), k) among equal maxima) instead of emulating
the shared-memory tree, and ball query is a vectorised mask + argsort.

Citations relative to /root/reference/pointnet2/_ext_src.
"""
import math

import numpy as np

f32 = np.float32


def fma32(a, b, c):
    """Correctly rounded fp32 fma for arrays: exact product/sum in float64
    (24+24 bit product is exact in 53 bits; the sum of that product with an
    fp32 addend can need more than 53 bits, so use the 2-sum trick)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    c = np.asarray(c, dtype=np.float64)
    p = a * b                     # exact
    s = p + c                     # rounded to f64
    # error of the f64 addition (exact, Knuth two-sum)
    bb = s - p
    err = (p - (s - bb)) + (c - bb)
    # round-to-odd emulation: nudge s toward the true value if inexact and
    # s is a tie candidate for f32 rounding.
    out = s.astype(np.float32)
    # fix double rounding: if s lies exactly half-way between two f32 and err != 0
    lo = out.astype(np.float64)
    half = (s - lo)
    nxt = np.where(half > 0, np.nextafter(out, f32(np.inf)), np.nextafter(out, f32(-np.inf))).astype(np.float64)
    is_tie = (np.abs(half) * 2 == np.abs(nxt - lo)) & (err != 0) & np.isfinite(s)
    # at a tie, astype rounded to even; the true value is s+err, so pick by sign of err
    toward_nxt = is_tie & (np.sign(err) == np.sign(half))
    toward_lo = is_tie & (np.sign(err) != np.sign(half))
    out = np.where(toward_nxt, nxt.astype(np.float32), out)
    out = np.where(toward_lo, lo.astype(np.float32), out)
    return out.astype(np.float32)


def sumsq3(a, b, c, mode=0):
    a = np.asarray(a, f32); b = np.asarray(b, f32); c = np.asarray(c, f32)
    if mode == 0:
        t = (b * b).astype(f32)
        t = fma32(a, a, t)
        t = fma32(c, c, t)
        return t
    return ((a * a + b * b).astype(f32) + (c * c).astype(f32)).astype(f32)


def opt_n_threads(w):
    """cuda_utils.h:20-24"""
    if w <= 0:
        return 1
    p = int(math.log(float(w)) / math.log(2.0))
    return max(min(1 << p, 512), 1)


def _bitrev(x, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


def fps(xyz, m, mode=0):
    """sampling_gpu.cu:74-178 for one scene, xyz (n,3) f32 -> (m,) i32."""
    xyz = np.asarray(xyz, f32)
    n = xyz.shape[0]
    out = np.zeros((m,), np.int32)
    if m <= 0 or n <= 0:
        return out
    bs = opt_n_threads(n)
    bits = bs.bit_length() - 1
    temp = np.full((n,), f32(1e10), f32)
    mag = sumsq3(xyz[:, 0], xyz[:, 1], xyz[:, 2], mode)
    valid = ~(mag.astype(np.float64) <= 1e-3)
    ks = np.arange(n)
    rev = np.array([_bitrev(int(k) % bs, bits) for k in ks], np.int64)
    old = 0
    for j in range(1, m):
        c = xyz[old]
        d = sumsq3(xyz[:, 0] - c[0], xyz[:, 1] - c[1], xyz[:, 2] - c[2], mode)
        temp = np.where(valid, np.minimum(d, temp), temp).astype(f32)
        if not valid.any():
            old = 0
        else:
            cand = np.where(valid, temp, f32(-1))
            mx = cand.max()
            tie = ks[valid & (cand == mx)]
            # a thread keeps its lowest k among equal values (strict >), threads
            # are merged by the tree: lowest bit-reversed tid wins
            key = rev[tie] * (n + 1) + tie
            old = int(tie[np.argmin(key)])
        out[j] = old
    return out


def ball_query(new_xyz, xyz, radius, nsample, mode=0):
    """ball_query_gpu.cu:14-49 for one scene."""
    new_xyz = np.asarray(new_xyz, f32); xyz = np.asarray(xyz, f32)
    m = new_xyz.shape[0]
    r2 = f32(f32(radius) * f32(radius))
    out = np.zeros((m, nsample), np.int32)
    for j in range(m):
        c = new_xyz[j]
        d2 = sumsq3(c[0] - xyz[:, 0], c[1] - xyz[:, 1], c[2] - xyz[:, 2], mode)
        hits = np.nonzero(d2 < r2)[0][:nsample]
        if len(hits):
            out[j, :] = hits[0]
            out[j, :len(hits)] = hits
    return out


def three_nn(unknown, known, mode=0):
    """interpolate_gpu.cu:14-64 for one scene: stable 3 smallest (ascending index on ties)."""
    unknown = np.asarray(unknown, f32); known = np.asarray(known, f32)
    n = unknown.shape[0]
    m = known.shape[0]
    d2 = np.zeros((n, 3), f32)
    idx = np.zeros((n, 3), np.int32)
    for j in range(n):
        u = unknown[j]
        d = sumsq3(u[0] - known[:, 0], u[1] - known[:, 1], u[2] - known[:, 2], mode)
        order = np.argsort(d, kind="stable")[:3]
        for t in range(3):
            if t < len(order):
                d2[j, t] = d[order[t]]
                idx[j, t] = order[t]
            else:
                d2[j, t] = np.float32(np.inf)      # float(1e40) -> +inf
                idx[j, t] = 0
    return d2, idx
