"""Scalar Philox4x32-10 on the host for the few draws the host control flow needs itself (the offset of the
systematic comb, the multinomial split of draws across ranks).  Same stream layout as csrc/common.h."""

_M0, _M1, _W0, _W1, _MASK = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85, 0xFFFFFFFF


def philox4x32(c0, c1, c2, c3, k0, k1):
    for _ in range(10):
        p0, p1 = _M0 * c0, _M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & _MASK, p1 & _MASK, ((p0 >> 32) ^ c3 ^ k1) & _MASK, p0 & _MASK
        k0, k1 = (k0 + _W0) & _MASK, (k1 + _W1) & _MASK
    return c0, c1, c2, c3


def uniform_scalar(seed, tick, tag, item=0, draw=0):
    r = philox4x32(item & _MASK, draw & _MASK, tick & _MASK, tag & _MASK, seed & _MASK, (seed >> 32) & _MASK)
    k = ((r[0] >> 5) << 26) | (r[1] >> 6)
    return k * 2.0 ** -53
