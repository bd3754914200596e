#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats` output) into a small
per-kernel CSV: name, calls, total_us, avg_us, pct — the same figures `--stats` prints.
    python tools/rocpd_summary.py gpurun_out/prof/x_results.db profiles/r01_kernel_stats.csv"""
import csv
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "pct"])
        for name, calls, tot, avg, pct in rows:
            w.writerow([name, calls, round(tot, 3), round(avg, 3), round(pct, 3)])  # view is already in microseconds
    print(f"{len(rows)} kernels -> {out}")


if __name__ == "__main__":
    main(*sys.argv[1:3])
