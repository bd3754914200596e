"""GPU occupancy over the steady state of an in-flight bench run, from a rocprofv3 --kernel-trace database:
    python tools/timeline_inflight.py results.db
Span between the 4th and the last head_conv (end of a batch), union of busy intervals, summed kernel time (= average number
of kernels running at once x span), and the largest idle gaps."""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = list(con.execute("select name, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "head_conv" in r[0]]
seg = rows[idx[3] + 1:idx[-1] + 1]
nb = len(idx) - 4
span = (max(r[2] for r in seg) - seg[0][1]) / 1e6
tot = sum(r[2] - r[1] for r in seg) / 1e6
busy, (cs, ce), gaps = 0, (seg[0][1], seg[0][2]), []
for name, s, e in seg[1:]:
    if s > ce:
        busy += ce - cs
        gaps.append((s - ce) / 1e3)
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print(f"{nb} batches, {len(seg)} kernels: span {span:.2f} ms ({span / nb:.3f} / batch), busy {busy / 1e6:.2f} ms, idle {span - busy / 1e6:.3f} ms "
      f"({100 * (1 - busy / 1e6 / span):.1f} %), kernel-time sum {tot:.2f} ms = {tot / span:.2f} kernels in flight on average")
gaps.sort(reverse=True)
print("largest gaps (us):", [round(g, 1) for g in gaps[:8]], " gaps > 5 us:", sum(1 for g in gaps if g > 5), " total", round(sum(gaps) / 1e3, 3), "ms")
