#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03k.txt; : > $O
timeout 2400 python -m pytest tests -q -m gpu -s 2>&1 | grep -E "passed|failed|\[server\]|Error|error" | tail -12 >> $O
for i in 1 2; do timeout 600 python bench.py --steps 24 --warmup 4 --min-seconds 1.0 --no-cpu-baseline --no-roofline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms in flight (", d["ms_per_step_min"], "..", d["ms_per_step_max"], "),", d.get("sequential_ms_per_step"), "one at a time")' >> $O; done
timeout 300 python tools/phase_breakdown.py --reps 4 2>/dev/null | grep -A8 "dec.s2\]\|dec.s1\]" >> $O
