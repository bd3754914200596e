#!/bin/bash
# r03b: second root-cause session: variant builds of the fused-prep kernel + the transcendental-forwarding microbenchmark
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03b.txt; : > $O
run() { echo "### $*" >> $O; timeout 300 env "$@" 2>&1 | grep -v amdgpu.ids >> $O; }
L=$PWD/smalltts_amd
echo "### tools/ubench/bin/trans_hazard" >> $O; timeout 300 tools/ubench/bin/trans_hazard >> $O 2>&1
run SMTTS_ATTN_PREP=1 python tools/stress_prep.py count 96
for v in 1 4 5 6 8 9 10 11 12; do
  run SMTTS_ATTN_PREP=1 SMTTS_LIB=$L/libdbg_v$v.so python tools/stress_prep.py count 96
done
