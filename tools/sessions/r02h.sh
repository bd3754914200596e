#!/bin/bash
O=gpurun_out/r02h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
L=smalltts_amd/libsmalltts_hip
R=3 timeout 1500 bash tools/ab_r02.sh $O "base|X=1|$L.so" "nw4x2|X=1|${L}_h.so" "nobarrier_probe|X=1|${L}_i.so" "up_g3_k1024|SMTTS_UP_G3_MINK=1024|$L.so" "up_g3_k512|SMTTS_UP_G3_MINK=512|$L.so"
SMTTS_LIB=$(realpath ${L}_h.so) timeout 300 python -m pytest tests/test_precision_gpu.py -q -m gpu -k codec > $O/tests_h.txt 2>&1
