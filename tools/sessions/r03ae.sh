#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03ae.txt; : > $O
bash tools/ab_envs.sh 3 "SMTTS_PERSIST_CUS=0" "SMTTS_X=1" >> $O 2>&1
timeout 900 python tools/stress_determinism.py 16 2>&1 | grep -v amdgpu >> $O
timeout 900 python -m pytest tests/test_api_gpu.py tests/test_server_gpu.py tests/test_bench_gpu.py -q -m gpu 2>&1 | tail -3 >> $O
for w in clone; do for e in 0 192; do printf "clone PERSIST=$e " >> $O; SMTTS_PERSIST_CUS=$e python bench.py --workload clone --steps 40 --no-cpu-baseline --no-roofline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("sequential_ms_per_step"))' >> $O; done; done
