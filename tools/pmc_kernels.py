"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel name:  python tools/pmc_kernels.py file.csv [substring]"""
import csv, collections, sys
rows = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = r["Kernel_Name"].split("(")[0][:70]
        rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for k, c in rows.items():
    if pat not in k:
        continue
    n = max(cnt[(k, name)] for name in c)
    print(f"{k}  (x{n})")
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    for name, v in sorted(c.items()):
        extra = f"  {100 * v / wc:5.1f} % of wave cycles" if wc and name.startswith("SQ_") and name != "SQ_WAVE_CYCLES" else ""
        print(f"    {name:28s} {v / n:16.0f} / launch{extra}")
