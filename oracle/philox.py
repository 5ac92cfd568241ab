"""Philox4x32-10 counter-based RNG, NumPy restatement of the device generator.

Test infrastructure (see ``oracle/__init__.py``).  The reference uses NumPy's
global MT19937 stream (tempest/mcmc.py:169,236,243,307; steps/mutate.py:102,129;
steps/resample.py:80; tools.py:217; modes.py:199,272), which cannot be
reproduced on a GPU.  The build replaces it by a counter-based
stream; this file is the CPU twin of ``tempest_amd/csrc/philox.h`` so device
kernels can be checked draw-for-draw.

Stream layout (shared with the device code):
    key     = (seed & 0xffffffff, seed >> 32)
    counter = (c0 = item index (particle / output slot),
               c1 = draw index inside (item, tick, tag),
               c2 = tick  (one per RNG-consuming launch, chosen by the host),
               c3 = tag   (purpose of the draw, TAG_* below))
One Philox call yields four 32-bit words = two 53-bit uniforms.
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK32 = np.uint64(0xFFFFFFFF)

TAG_PRIOR = 1      # u ~ U(0,1)^d at beta=0            (steps/mutate.py:102)
TAG_NORMAL = 2     # z ~ N(0,I) of the proposals        (mcmc.py:243,307)
TAG_GAMMA = 3      # Gamma draw of tpCN                 (mcmc.py:236)
TAG_ACCEPT = 4     # Metropolis uniform                 (mcmc.py:169)
TAG_RESAMPLE = 5   # multinomial resampling             (steps/resample.py:80)
TAG_UPSAMPLE = 6   # x4 up-sampling before fit_mvstud   (modes.py:199,272)
TAG_REPAIR = 7     # inf-likelihood repair              (steps/mutate.py:129)
TAG_SYST = 8       # systematic resampling offset U     (tools.py:217)
TAG_CLUSTER = 9    # k-means++ / GMM initialisation     (cluster.py:144,157)

TWO_M53 = 2.0 ** -53


def philox4x32(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10.  Inputs broadcast; returns 4 uint32 arrays."""
    c0, c1, c2, c3 = np.broadcast_arrays(
        np.asarray(c0, dtype=np.uint64), np.asarray(c1, dtype=np.uint64),
        np.asarray(c2, dtype=np.uint64), np.asarray(c3, dtype=np.uint64))
    c0 = c0 & MASK32; c1 = c1 & MASK32; c2 = c2 & MASK32; c3 = c3 & MASK32
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0 = p0 >> np.uint64(32); lo0 = p0 & MASK32
        hi1 = p1 >> np.uint64(32); lo1 = p1 & MASK32
        n0 = hi1 ^ c1 ^ np.uint64(k0)
        n2 = hi0 ^ c3 ^ np.uint64(k1)
        c0, c1, c2, c3 = n0, lo1, n2, lo0
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return (c0.astype(np.uint32), c1.astype(np.uint32),
            c2.astype(np.uint32), c3.astype(np.uint32))


def _k53(hi, lo):
    """53-bit integer from two 32-bit words: 27 high bits of hi, 26 of lo."""
    return ((hi.astype(np.uint64) >> np.uint64(5)) << np.uint64(26)) | (
        lo.astype(np.uint64) >> np.uint64(6))


def uniform_pair(seed, item, draw, tick, tag):
    """Two U[0,1) doubles per (item, draw, tick, tag): k * 2^-53."""
    seed = int(seed)
    r0, r1, r2, r3 = philox4x32(item, draw, tick, tag, seed & 0xFFFFFFFF, seed >> 32)
    return _k53(r0, r1).astype(np.float64) * TWO_M53, _k53(r2, r3).astype(np.float64) * TWO_M53


def normal_pair(seed, item, draw, tick, tag):
    """Two N(0,1) doubles by Box-Muller: r = sqrt(-2 ln u1), u1 in (0,1]."""
    seed = int(seed)
    r0, r1, r2, r3 = philox4x32(item, draw, tick, tag, seed & 0xFFFFFFFF, seed >> 32)
    u1 = (_k53(r0, r1).astype(np.float64) + 1.0) * TWO_M53
    u2 = _k53(r2, r3).astype(np.float64) * TWO_M53
    r = np.sqrt(-2.0 * np.log(u1))
    th = 2.0 * np.pi * u2
    return r * np.cos(th), r * np.sin(th)


def normals(seed, items, n_dim, tick, tag=TAG_NORMAL, attempt=0):
    """(len(items), n_dim) standard normals; pair p of attempt a is draw a*ceil(d/2)+p."""
    items = np.asarray(items, dtype=np.uint64)
    npairs = (n_dim + 1) // 2
    attempt = np.asarray(attempt, dtype=np.uint64)
    out = np.empty((items.size, n_dim))
    for p in range(npairs):
        z0, z1 = normal_pair(seed, items, attempt * np.uint64(npairs) + np.uint64(p), tick, tag)
        out[:, 2 * p] = z0
        if 2 * p + 1 < n_dim:
            out[:, 2 * p + 1] = z1
    return out


def uniforms(seed, items, n_dim, tick, tag=TAG_PRIOR):
    """(len(items), n_dim) U[0,1) doubles; pair p is draw p."""
    items = np.asarray(items, dtype=np.uint64)
    out = np.empty((items.size, n_dim))
    for p in range((n_dim + 1) // 2):
        a, b = uniform_pair(seed, items, p, tick, tag)
        out[:, 2 * p] = a
        if 2 * p + 1 < n_dim:
            out[:, 2 * p + 1] = b
    return out


def uniform1(seed, items, tick, tag, draw=0):
    """One U[0,1) per item (first of the pair)."""
    return uniform_pair(seed, np.asarray(items, dtype=np.uint64), draw, tick, tag)[0]


def gamma_candidate(seed, items, att, tick, tag=TAG_GAMMA):
    """The candidate of one Marsaglia-Tsang attempt from ONE Philox call (device: tph_rng::gamma_candidate): a normal
    x = sqrt(-2 ln u1) cos(2 pi u2) with a 53-bit u1 in (0,1] and a 32-bit angle u2, and log of a (0,1] uniform of 32 bits."""
    seed = int(seed)
    r0, r1, r2, r3 = philox4x32(items, att, tick, tag, seed & 0xFFFFFFFF, seed >> 32)
    u1 = (_k53(r0, r1).astype(np.float64) + 1.0) * TWO_M53
    x = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * (r2.astype(np.float64) * 2.0 ** -32))
    logu = np.log((r3.astype(np.float64) + 1.0) * 2.0 ** -32)
    return x, logu


def gamma_mt(seed, items, shape, tick, tag=TAG_GAMMA, max_attempts=64):
    """Gamma(shape, 1) by Marsaglia-Tsang; attempt a uses draw a (one Philox call for its normal and its uniform).

    shape < 1 is boosted: G(a) = G(a+1) * U^(1/a) with U from draw 2*max_attempts.
    Restates what np.random.gamma supplies to mcmc.py:236 (same law, different stream).
    """
    items = np.asarray(items, dtype=np.uint64)
    shape = np.broadcast_to(np.asarray(shape, dtype=np.float64), items.shape).copy()
    boost = shape < 1.0
    a = np.where(boost, shape + 1.0, shape)
    d = a - 1.0 / 3.0
    c = 1.0 / np.sqrt(9.0 * d)
    out = np.full(items.shape, np.nan)
    todo = np.ones(items.shape, dtype=bool)
    for att in range(max_attempts):
        if not todo.any():
            break
        x, logu = gamma_candidate(seed, items, att, tick, tag)
        v = 1.0 + c * x
        v = v * v * v
        with np.errstate(invalid="ignore", divide="ignore"):
            ok = (v > 0.0) & (logu < 0.5 * x * x + d - d * v + d * np.log(v))
        take = todo & ok
        out[take] = (d * v)[take]
        todo &= ~ok
    out[todo] = d[todo]  # pathological: fall back to the mode-ish value
    if boost.any():
        ub, _ = uniform_pair(seed, items, 2 * max_attempts, tick, tag)
        ub = ub + TWO_M53
        out = np.where(boost, out * ub ** (1.0 / shape), out)
    return out
