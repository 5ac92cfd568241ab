"""x4 up-sampling counts (tph_multinomial_counts): draw-order lookups against the sorted draws counted by the owners of the cdf's
tiles, over history sizes and draw counts -- where does the sorted path start to pay (MC_SORT_MIN in csrc/resample.hip)?
One JSON line per (rows, draws): median microseconds of either path (HIP events around the call, host read of the count included)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from tempest_amd.device import HipContext, OPT_SORTED_DRAWS
    dev = torch.device("cuda", 0)
    c = HipContext(4)
    c.use_current_stream()
    rs = np.random.RandomState(1)
    for rows in (1 << 20, 3 << 20, 13 << 20, 26 << 20):
        w = np.exp(2.0 * rs.randn(rows))
        w[rs.rand(rows) < 0.5] = 0.0
        w /= w.sum()
        cdf = c.cdf(torch.from_numpy(w).to(dev))
        for draws in (1 << 19, 1 << 20, 1 << 21, 1 << 22, 1 << 23, 1 << 24):
            if draws > 4 * rows:
                continue
            rec = {"rows": rows, "draws": draws}
            for name, mode in (("lookups_us", 0), ("sorted_us", 2)):
                c.set_option(OPT_SORTED_DRAWS, mode)
                ts = []
                for rep in range(7):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    out = c.multinomial_counts(cdf, seed=3, tick=rep, kept_count=None, factor=1, n_draw_max=draws)
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3)
                rec[name] = round(float(np.median(ts[2:])), 1)
                rec[name.replace("_us", "_sum")] = int(out.sum().item())
            c.set_option(OPT_SORTED_DRAWS, 1)
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
