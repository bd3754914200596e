#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03p.txt; : > $O
L=$PWD/smalltts_amd
SMTTS_LIB=$L/libtimeline.so timeout 600 python tools/gemm3_timeline.py 2>&1 | grep -v amdgpu >> $O
bash tools/ab_envs.sh 4 "SMTTS_LIB=$L/libnolate.so" "SMTTS_LIB=$L/libsmalltts_hip.so" >> $O 2>&1
