#!/bin/bash
O=gpurun_out/r02o; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_dit_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
L=smalltts_amd/libsmalltts_hip
R=3 timeout 1500 bash tools/ab_r02.sh $O "fused_prep|SMTTS_ATTN_PREP=1|$L.so" "qk_prep_launch|SMTTS_ATTN_PREP=0|$L.so"
