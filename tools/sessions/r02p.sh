#!/bin/bash
O=gpurun_out/r02p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -c "
import torch
print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')
for p in (-2,-1,0,1,2):
    try:
        s=torch.cuda.Stream(priority=p); print(p, '->', s.priority)
    except Exception as e: print(p, 'err', e)
" > $O/prio.txt 2>&1
L=smalltts_amd/libsmalltts_hip
for i in 1 2 3; do
 for v in "base|" "front0|SMTTS_BENCH_FRONT_PRIO=0" "front_hi|SMTTS_BENCH_FRONT_PRIO=-1"; do
  IFS='|' read -r label envs <<< "$v"
  printf "%-12s " "$label" >> $O/ab.txt
  env $envs timeout 300 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms in flight,", d.get("sequential_ms_per_step"), "one at a time")' >> $O/ab.txt
 done
done
for n in 2 4 6; do
  printf "front_hi in_flight=%s " $n >> $O/ab.txt
  SMTTS_BENCH_FRONT_PRIO=-1 timeout 300 python bench.py --steps 60 --warmup 6 --in-flight $n --no-cpu-baseline --no-roofline 2>/dev/null | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms in flight")' >> $O/ab.txt
done
