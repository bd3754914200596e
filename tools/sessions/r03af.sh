#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03af.txt; : > $O
bash tools/ab_envs.sh 3 "SMTTS_PERSIST_MASK=7" "SMTTS_PERSIST_MASK=1" "SMTTS_PERSIST_MASK=3" "SMTTS_PERSIST_MASK=5" "SMTTS_PERSIST_MASK=6" >> $O 2>&1
for k in 2 3 4 5 6; do
  printf "in_flight=%s  " $k >> $O
  timeout 300 python bench.py --steps 60 --warmup 6 --in-flight $k --no-cpu-baseline --no-roofline 2>/dev/null | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms/step,", d["value"], "audio-s/s; sequential", d.get("sequential_ms_per_step"))' >> $O
done
