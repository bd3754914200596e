#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03r.txt; : > $O
L=$PWD/smalltts_amd
bash tools/ab_envs.sh 3 "SMTTS_LIB=$L/libprev.so" "SMTTS_LIB=$L/libepi1.so" "SMTTS_LIB=$L/libepi2.so" "SMTTS_LIB=$L/libsmalltts_hip.so" >> $O 2>&1
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -4 >> $O
timeout 300 python tools/phase_breakdown.py --reps 4 2>/dev/null >> $O
