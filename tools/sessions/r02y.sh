#!/bin/bash
O=gpurun_out/r02y; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_codec_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
L=smalltts_amd/libsmalltts_hip
R=3 timeout 1500 bash tools/ab_r02.sh $O "desync|SMTTS_FS_DESYNC_US=-1|$L.so" "off|SMTTS_FS_DESYNC_US=0|$L.so" "d25|SMTTS_FS_DESYNC_US=25|$L.so" "d80|SMTTS_FS_DESYNC_US=80|$L.so"
