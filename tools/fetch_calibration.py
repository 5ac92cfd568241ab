"""FETCH_SIZE against known byte counts (tools/ubench_fetch.hip): reads the probe's expected figures and the rocprofv3 counter /
trace CSVs of a run of it, prints one JSON object: per access pattern the counter (KB x 1024), the bytes used, the 64-byte sectors
and 128-byte lines touched, and the ratios that tell which of them the counter tallies.
  python3 tools/fetch_calibration.py <expected.json> <rocprof dir>"""
import json
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from roofline_table import _read      # noqa: E402


def main(expected, base):
    exp = json.load(open(expected))
    dur, fetch = _read(base, "FETCH_SIZE")
    out = {"what": "rocprofv3 --pmc FETCH_SIZE per access pattern over a 2 GiB table (tools/ubench_fetch.hip), second launch of each",
           "patterns": {}}
    for name, e in exp.items():
        if not isinstance(e, dict):
            continue
        k = next((n for n in fetch if n.startswith(name) or n.startswith("void " + name)), None)
        if k is None:
            continue
        f = fetch[k][-1] * 1024.0
        t = dur[k][-1] / 1e3 if k in dur else None
        row = {"FETCH_SIZE_bytes": f, "duration_us": t}
        row.update(e)
        for key in ("bytes", "bytes_used", "sectors64", "sectors64_expected", "lines128", "lines128_expected"):
            if key in e:
                unit = 64.0 if key.startswith("sectors") else 128.0 if key.startswith("lines") else 1.0
                row["counter_over_" + key] = round(f / (e[key] * unit), 4)
        out["patterns"][name] = row
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
