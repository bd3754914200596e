#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03l.txt; : > $O
bash tools/ab_envs.sh 3 "SMTTS_QKV_BIG_MINM=641" "SMTTS_QKV_BIG_MINM=512" >> $O 2>&1
