#!/bin/bash
O=gpurun_out/r02z; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_codec_gpu.py tests/test_precision_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
SMTTS_LIB=$(realpath smalltts_amd/libsmalltts_hip_xpre.so) timeout 600 python -m pytest tests/test_codec_gpu.py -x -q -m gpu > $O/tests_xpre.txt 2>&1; echo "rc=$?" >> $O/tests_xpre.txt
L=smalltts_amd/libsmalltts_hip
R=3 timeout 1500 bash tools/ab_r02.sh $O "hoist|X=1|$L.so" "old|X=1|${L}_old.so" "xpre|X=1|${L}_xpre.so"
