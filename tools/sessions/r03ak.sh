#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03ak.txt; : > $O
for i in 1 2; do for e in "SMTTS_GEMM_DEEP_TP=0 SMTTS_PERSIST_CUS=0" "SMTTS_GEMM_DEEP_TP=1 SMTTS_PERSIST_CUS=0" "SMTTS_GEMM_DEEP_TP=1 SMTTS_PERSIST_CUS=192"; do
  printf "%-50s teacher128 " "$e" >> $O
  env $e timeout 400 python bench.py --workload teacher128 --steps 10 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("sequential_ms_per_step"))' >> $O
  printf "%-50s clone      " "$e" >> $O
  env $e timeout 400 python bench.py --workload clone --steps 60 --no-cpu-baseline --no-roofline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("sequential_ms_per_step"))' >> $O
done; done
