#!/bin/bash
O=gpurun_out/r02aq; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_precision_gpu.py tests/test_dit_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -s > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
SMTTS_GEMM_DEEP=1 timeout 300 python tools/gemm_cfg_sweep.py 2>/dev/null | grep -E "ff1|dit.qkvg  " > $O/sweep.txt
for i in 1 2; do
  printf "teacher128  " >> $O/teacher.txt
  timeout 400 python bench.py --workload teacher128 --steps 8 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms in flight,", d.get("sequential_ms_per_step"), "one at a time")' >> $O/teacher.txt
done
timeout 300 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c '
import sys,json
d=json.loads(sys.stdin.read()); print("dmd4", d["ms_per_step"], d.get("sequential_ms_per_step"))' >> $O/teacher.txt
