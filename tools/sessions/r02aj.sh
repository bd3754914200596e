#!/bin/bash
O=gpurun_out/r02aj; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
L=smalltts_amd/libsmalltts_hip
R=3 timeout 1200 bash tools/ab_r02.sh $O "default|SMTTS_ATTN_RES=1|$L.so" "res64|SMTTS_ATTN_RES=64|$L.so"
