#!/bin/bash
O=gpurun_out/r02j; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
L=smalltts_amd/libsmalltts_hip
R=3 timeout 1200 bash tools/ab_r02.sh $O "gelu_q5|X=1|$L.so" "gelu_as3|X=1|${L}_b.so"
timeout 600 python -m pytest tests/test_precision_gpu.py tests/test_codec_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
