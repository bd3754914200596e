#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03m.txt; : > $O
timeout 1500 python -m pytest tests/test_dit_gpu.py tests/test_api_gpu.py tests/test_precision_gpu.py -x -q 2>&1 | tail -4 >> $O
bash tools/ab_envs.sh 3 "SMTTS_SINGLE_STREAM=0" "SMTTS_SINGLE_STREAM=1" >> $O 2>&1
