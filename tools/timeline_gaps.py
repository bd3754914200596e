"""Idle time inside one steady-state batch from a rocprofv3 --kernel-trace database:  python tools/timeline_gaps.py results.db"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = list(con.execute("select name, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "head_conv" in r[0]]   # last kernel of a decode = end of a batch
for b in range(len(idx) - 4, len(idx)):
    seg = rows[idx[b - 1] + 1:idx[b] + 1]
    span = (seg[-1][2] - seg[0][1]) / 1e6
    tot = sum(r[2] - r[1] for r in seg) / 1e6
    busy, (cs, ce) = 0, (seg[0][1], seg[0][2])
    big = []
    for i in range(1, len(seg)):
        s, e = seg[i][1], seg[i][2]
        if s > ce:
            busy += ce - cs
            if s - ce > 20000:
                big.append(((s - ce) / 1e3, seg[i - 1][0][:40], seg[i][0][:40]))
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    print(f"batch {b}: {len(seg)} kernels, span {span:.3f} ms, kernel-time sum {tot:.3f}, busy {busy / 1e6:.3f}, idle {span - busy / 1e6:.3f} ms")
    for g in big[:6]:
        print(f"     gap {g[0]:6.1f} us between {g[1]} and {g[2]}")
