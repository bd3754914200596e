#!/bin/bash
O=gpurun_out/r02q; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
 for v in "base|" "base_q8|GPU_MAX_HW_QUEUES=8" "front0_q8|SMTTS_BENCH_FRONT_PRIO=0 GPU_MAX_HW_QUEUES=8" "front_hi_q8|SMTTS_BENCH_FRONT_PRIO=-1 GPU_MAX_HW_QUEUES=8" "base_q2|GPU_MAX_HW_QUEUES=2"; do
  IFS='|' read -r label envs <<< "$v"
  printf "%-12s " "$label" >> $O/ab.txt
  env $envs timeout 300 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms in flight,", d.get("sequential_ms_per_step"), "one at a time")' >> $O/ab.txt
 done
done
for n in 4 6; do
  printf "base_q8 in_flight=%s " $n >> $O/ab.txt
  GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --steps 60 --warmup 6 --in-flight $n --no-cpu-baseline --no-roofline 2>/dev/null | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms in flight")' >> $O/ab.txt
done
