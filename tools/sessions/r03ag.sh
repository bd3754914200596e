#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03ag.txt; : > $O
bash tools/ab_envs.sh 3 "SMTTS_GEMM_DEEP_TP=0" "SMTTS_GEMM_DEEP_TP=1" "SMTTS_GEMM_DEEP_TP=0 SMTTS_KSPLIT_TP=1" >> $O 2>&1
