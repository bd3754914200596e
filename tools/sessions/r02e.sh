#!/bin/bash
O=gpurun_out/r02e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 python tools/debug_weights.py > $O/debug_weights.txt 2>&1
L=smalltts_amd/libsmalltts_hip
R=3 timeout 1500 bash tools/ab_r02.sh $O "base|X=1|$L.so" "gelu5|X=1|${L}_b.so" "pfd3|X=1|${L}_c.so" "s8|X=1|${L}_d.so" "pfd3_s8|X=1|${L}_e.so"
timeout 600 python -m pytest tests/test_precision_gpu.py tests/test_codec_gpu.py -x -q -m gpu -s > $O/tests_prec.txt 2>&1; echo "rc=$?" >> $O/tests_prec.txt
