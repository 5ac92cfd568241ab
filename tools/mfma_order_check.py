"""Which scalar arithmetic reproduces v_mfma_f64_16x16x4_f64 bit for bit?  Reads the dump of tools/ubench_mfma_order.hip and tests
hypotheses in exact rational arithmetic (float(Fraction) rounds to nearest even)."""
import itertools
import json
import struct
import sys
from fractions import Fraction as F

import numpy as np


def fma(a, b, c):
    return float(F(a) * F(b) + F(c))


def main(path):
    raw = open(path, "rb").read()
    trials = struct.unpack("i", raw[:4])[0]
    body = np.frombuffer(raw[8:], dtype=np.float64)
    na, nc = trials * 64, trials * 256
    A = body[:na].reshape(trials, 16, 4)
    B = body[na:2 * na].reshape(trials, 4, 16)
    C = body[2 * na:2 * na + nc].reshape(trials, 16, 16)
    D = body[2 * na + nc:].reshape(trials, 16, 16)
    hyps = {}
    for perm in itertools.permutations(range(4)):
        def chain(a, b, c, perm=perm):
            t = c
            for k in perm:
                t = fma(a[k], b[k], t)
            return t
        hyps["fma chain k=%s" % "".join(map(str, perm))] = chain
    hyps["exact sum, one rounding"] = lambda a, b, c: float(sum((F(a[k]) * F(b[k]) for k in range(4)), F(c)))
    hyps["products rounded, added to c in k order"] = lambda a, b, c: ((((c + a[0] * b[0]) + a[1] * b[1]) + a[2] * b[2]) + a[3] * b[3])

    def pairs(a, b, c):
        p = fma(a[1], b[1], a[0] * b[0])
        q = fma(a[3], b[3], a[2] * b[2])
        return (c + p) + q
    hyps["pairs (fma(a1b1, a0b0)) + (..) + c"] = pairs
    score = {h: 0 for h in hyps}
    total = 0
    for t in range(trials):
        for i in range(16):
            for j in range(16):
                a, b, c, d = A[t, i], B[t, :, j], float(C[t, i, j]), float(D[t, i, j])
                total += 1
                for h, fn in hyps.items():
                    if fn([float(v) for v in a], [float(v) for v in b], c) == d:
                        score[h] += 1
    best = sorted(score.items(), key=lambda kv: -kv[1])[:6]
    print(json.dumps({"entries": total, "matches": dict(best)}))


if __name__ == "__main__":
    main(sys.argv[1])
