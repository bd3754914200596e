#!/bin/bash
O=gpurun_out/r02m; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_codec_gpu.py tests/test_precision_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -s > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
L=smalltts_amd/libsmalltts_hip
R=3 timeout 1200 bash tools/ab_r02.sh $O "block_wave|SMTTS_BLOCK_WAVE=1|$L.so" "two_kernels|SMTTS_BLOCK_WAVE=0|$L.so"
