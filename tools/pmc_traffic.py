#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; units of KiB... see below).
MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE reports exactly half the bytes of a wide coalesced streaming read, so
the read side is doubled; WRITE_SIZE is taken as reported (uncalibrated).  bytes = value * 1024.
    python tools/pmc_traffic.py fetch.csv write.csv out.json"""
import collections
import csv
import json
import re
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_names import prof_name  # noqa: E402


def short(name):
    m = re.match(r"(?:void )?([A-Za-z0-9_]+)(<.*>)?\(", name)
    return (m.group(1) + (m.group(2) or "")) if m else name


def load(path, counter):
    agg = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return agg


def main(fetch_csv, write_csv, out, sha=None, tag=None):
    fe, wr = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fe) | set(wr)):
        f = sum(fe.get(k, [0])) / max(1, len(fe.get(k, [0])))
        w = sum(wr.get(k, [0])) / max(1, len(wr.get(k, [0])))
        res[k] = {"launches": len(fe.get(k, [])), "fetch_kib_raw": round(f, 1), "write_kib_raw": round(w, 1),
                  "hbm_bytes_per_launch": round((2.0 * f + w) * 1024)}
    # the same figures keyed by bench.py's profiler names (launch-weighted over the template instantiations of a class)
    by = collections.defaultdict(lambda: [0.0, 0])
    raw = {}
    for path, counter in ((fetch_csv, "FETCH_SIZE"),):
        with open(path) as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] == counter:
                    raw.setdefault(short(r["Kernel_Name"]), r["Kernel_Name"])
    for k, v in res.items():
        n = prof_name(raw.get(k, k))
        if n and v["launches"]:
            by[n][0] += v["hbm_bytes_per_launch"] * v["launches"]
            by[n][1] += v["launches"]
    by_prof = {n: round(b / c) for n, (b, c) in by.items() if c}
    with open(out, "w") as fo:
        json.dump({"note": "per launch; read side doubled per the gfx950 FETCH_SIZE correction", "kernel_src_sha": sha, "tag": tag,
                   "kernels": res, "by_prof_name": by_prof}, fo, indent=1)
    top = sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:8]
    for k, v in top:
        print(f"{v['hbm_bytes_per_launch'] / 1e6:10.1f} MB/launch x{v['launches']:4d}  {k[:110]}")


if __name__ == "__main__":
    main(*sys.argv[1:6])
