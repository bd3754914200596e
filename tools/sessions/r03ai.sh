#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03ai.txt; : > $O
bash tools/ab_envs.sh 3 "SMTTS_DUAL_TP=0" "SMTTS_DUAL_TP=1" >> $O 2>&1
timeout 900 python -m pytest tests/test_api_gpu.py tests/test_server_gpu.py -q -m gpu 2>&1 | tail -3 >> $O
