"""Merge reference-ensemble result files (make_ref_ensembles.py --out ...) into tests/golden/ref_ensembles.json.
TEST INFRASTRUCTURE ONLY.  Usage: python3 oracle/merge_ref_ensembles.py extra.json [...]"""
import json
import sys

import numpy as np

MAIN = "/root/repo/tests/golden/ref_ensembles.json"


def main():
    out = json.load(open(MAIN))
    have = {(r["config"], r["seed"]) for r in out["runs"]}
    for path in sys.argv[1:]:
        for r in json.load(open(path))["runs"]:
            if (r["config"], r["seed"]) not in have:
                out["runs"].append(r)
                have.add((r["config"], r["seed"]))
    summ = {}
    for c in sorted({r["config"] for r in out["runs"]}):
        lz = np.array([r["logz"] for r in out["runs"] if r["config"] == c])
        summ[c] = dict(n=int(lz.size), logz_mean=float(lz.mean()), logz_std=float(lz.std(ddof=1)) if lz.size > 1 else None)
    out["summary"] = summ
    json.dump(out, open(MAIN, "w"), indent=0)
    print(json.dumps(summ, indent=1))


if __name__ == "__main__":
    main()
