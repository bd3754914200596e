#!/bin/bash
O=gpurun_out/r02k; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
L=smalltts_amd/libsmalltts_hip
R=3 timeout 1200 bash tools/ab_r02.sh $O "keepx|X=1|$L.so" "reread|X=1|${L}_b.so" "keepx_w4|SMTTS_GEMM_W4_MINM=2048|$L.so"
timeout 600 python -m pytest tests/test_precision_gpu.py tests/test_codec_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
