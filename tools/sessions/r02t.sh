#!/bin/bash
O=gpurun_out/r02t; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_codec_gpu.py tests/test_fullsize_gpu.py tests/test_api_gpu.py tests/test_kernels_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
for i in 1 2 3; do for v in 1 0; do
  printf "clone SMTTS_SMALLM_SPLITK=%s  " $v >> $O/clone.txt
  SMTTS_SMALLM_SPLITK=$v timeout 300 python bench.py --workload clone --steps 60 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms in flight,", d.get("sequential_ms_per_step"), "one at a time")' >> $O/clone.txt
done; done
timeout 300 python tools/phase_breakdown.py --reps 4 --workload clone > $O/phases_clone.txt 2>/dev/null
