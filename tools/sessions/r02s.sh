#!/bin/bash
O=gpurun_out/r02s; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2; do for v in 1 0 2; do
  printf "teacher128 SMTTS_ATTN_PREP=%s  " $v >> $O/teacher.txt
  SMTTS_ATTN_PREP=$v timeout 400 python bench.py --workload teacher128 --steps 8 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms in flight,", d.get("sequential_ms_per_step"), "one at a time")' >> $O/teacher.txt
done; done
timeout 300 python tools/phase_breakdown.py --reps 4 --workload clone > $O/phases_clone.txt 2>/dev/null
