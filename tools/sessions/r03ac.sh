#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03ac.txt; : > $O
bash tools/ab_envs.sh 3 "SMTTS_PERSIST_CUS=0" "SMTTS_PERSIST_CUS=240" "SMTTS_PERSIST_CUS=224" "SMTTS_PERSIST_CUS=192" >> $O 2>&1
