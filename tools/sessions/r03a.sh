#!/bin/bash
# r03a: root-cause session for the fused q / k prep attention kernel
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03a.txt; : > $O
run() { echo "### $*" >> $O; timeout 300 env "$@" >> $O 2>&1; }
L=$PWD/smalltts_amd
run SMTTS_ATTN_PREP=0 python tools/stress_prep.py count 48
run SMTTS_ATTN_PREP=1 python tools/stress_prep.py dump 48
run SMTTS_ATTN_PREP=1 python tools/stress_prep.py count 48
run SMTTS_ATTN_PREP=1 GPU_MAX_HW_QUEUES=1 python tools/stress_prep.py count 48
run SMTTS_ATTN_PREP=1 GPU_MAX_HW_QUEUES=2 python tools/stress_prep.py count 48
run SMTTS_ATTN_PREP=1 SMTTS_DBG_ATTN_LDS=163840 python tools/stress_prep.py count 48
for v in 1 2 3 4 5 6 7; do
  run SMTTS_ATTN_PREP=1 SMTTS_LIB=$L/libdbg_v$v.so python tools/stress_prep.py count 48
done
run SMTTS_ATTN_PREP=1 python tools/stress_prep.py neigh 32
run SMTTS_ATTN_PREP=0 python tools/stress_prep.py neigh 16
