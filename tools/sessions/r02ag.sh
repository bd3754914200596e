#!/bin/bash
O=gpurun_out/r02ag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_dit_gpu.py tests/test_api_gpu.py tests/test_precision_gpu.py tests/test_kernels_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
L=smalltts_amd/libsmalltts_hip
R=3 timeout 1200 bash tools/ab_r02.sh $O "new|X=1|$L.so" "old|X=1|${L}_old.so"
for i in 1 2; do for v in "" _old; do
  printf "teacher128 lib%s  " "$v" >> $O/teacher.txt
  SMTTS_LIB=$(realpath $L$v.so) timeout 400 python bench.py --workload teacher128 --steps 8 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms in flight,", d.get("sequential_ms_per_step"), "one at a time")' >> $O/teacher.txt
done; done
