#!/bin/bash
O=gpurun_out/r02ao; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_dit_gpu.py tests/test_precision_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
L=smalltts_amd/libsmalltts_hip
R=3 timeout 1200 bash tools/ab_r02.sh $O "by_group|SMTTS_CONVPOS_BY_GROUP=1|$L.so" "per_utt|SMTTS_CONVPOS_BY_GROUP=0|$L.so"
for i in 1 2; do for v in 1 0; do
  printf "teacher128 SMTTS_CONVPOS_BY_GROUP=%s  " $v >> $O/teacher.txt
  SMTTS_CONVPOS_BY_GROUP=$v timeout 400 python bench.py --workload teacher128 --steps 8 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms in flight,", d.get("sequential_ms_per_step"), "one at a time")' >> $O/teacher.txt
done; done
