#!/bin/bash
O=gpurun_out/r02ak; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_dit_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
for i in 1 2; do for v in 1 0; do
  printf "teacher128 SMTTS_ATTN_RES=%s  " $v >> $O/teacher.txt
  SMTTS_ATTN_RES=$v timeout 400 python bench.py --workload teacher128 --steps 8 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms in flight,", d.get("sequential_ms_per_step"), "one at a time")' >> $O/teacher.txt
done; done
timeout 600 python tools/phase_breakdown.py --workload teacher128 --reps 1 2>/dev/null | head -14 > $O/phases_teacher.txt
