#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03c.txt; : > $O
run() { echo "### $*" >> $O; timeout 300 env "$@" 2>&1 | grep -v amdgpu.ids >> $O; }
L=$PWD/smalltts_amd
run SMTTS_ATTN_PREP=1 python tools/stress_prep.py dump 96
for v in 7 14 15; do
  run SMTTS_ATTN_PREP=1 SMTTS_LIB=$L/libdbg_v$v.so python tools/stress_prep.py count 96
done
run SMTTS_ATTN_PREP=1 SMTTS_LIB=$L/libdbg_v14.so python tools/stress_prep.py dump 96
