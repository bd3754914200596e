#!/bin/bash
O=gpurun_out/r02am; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
L=smalltts_amd/libsmalltts_hip
R=3 timeout 1500 bash tools/ab_r02.sh $O "split_both|X=1|$L.so" "out_unsplit|SMTTS_KSPLIT_OUT=1|$L.so" "ff2_unsplit|SMTTS_KSPLIT_FF2=1|$L.so"
