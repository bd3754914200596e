#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03d.txt; : > $O
run() { echo "### $*" >> $O; timeout 600 env "$@" 2>&1 | grep -v amdgpu.ids >> $O; }
run SMTTS_ATTN_PREP=1 python tools/stress_prep.py aggr 32
echo "### kernel tests" >> $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" 2>&1 | tail -15 >> $O
timeout 1200 python -m pytest tests/test_dit_gpu.py tests/test_precision_gpu.py -x -q 2>&1 | tail -15 >> $O
