"""Generate tests/golden/ref_hgm_scale.json: what the REFERENCE's hierarchical mixture (tempest/cluster.py:420-600) decides on
clean four-mode sets of growing size -- in particular that it does NOT split 100 000 well-separated rows (the two-component
EM settles on a worse optimum than one Gaussian), which is what the device clustering reproduces at config scale.

TEST INFRASTRUCTURE ONLY.  Usage (scratch cwd; the reference is read-only and never travels):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference:/root/repo python3 /root/repo/oracle/make_ref_hgm_scale.py

Stored per set: the generator's arguments (the sets are re-created from RandomState(0) in the test), the number of clusters, and
the root cluster's BIC improvement and threshold as the reference prints them (parsed from its verbose output)."""
import contextlib
import io
import json
import os
import re

import numpy as np

SETS = ((8, 2000, 0.25, 0.025), (32, 4000, 0.2, 0.0125), (32, 4000, 0.3, 0.015), (32, 20000, 0.3, 0.015), (32, 100000, 0.3, 0.015))


def make_sets():
    """The five sets, drawn one after the other from RandomState(0) (tests/test_cluster_gpu.py draws them the same way)."""
    rs = np.random.RandomState(0)
    for d, n, sep, sig in SETS:
        mus = np.full((4, d), 0.5)
        for k, (a, b) in enumerate([(-1, -1), (-1, 1), (1, -1), (1, 1)]):
            mus[k, 0] += a * sep
            mus[k, 1] += b * sep
        lab = rs.randint(4, size=n)
        yield (d, n, sep, sig), mus[lab] + sig * rs.randn(n, d)


def main():
    from tempest.cluster import HierarchicalGaussianMixture
    out = []
    for (d, n, sep, sig), X in make_sets():
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            h = HierarchicalGaussianMixture(verbose=True, normalize=True)
            h.fit(X, np.ones(n) / n)
        m = re.search(r"Cluster 0: parent BIC=([-\d.]+), children BIC=([-\d.]+), improvement=([-\d.]+), threshold=([-\d.]+)", buf.getvalue())
        out.append(dict(n_dim=d, rows=n, sep=sep, sig=sig, K=int(h.n_clusters_), root_improvement=float(m.group(3)),
                        root_threshold=float(m.group(4))))
        print(out[-1], flush=True)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "ref_hgm_scale.json")
    json.dump({"reference": "minaskar/tempest 0.2.1", "sets": out}, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
