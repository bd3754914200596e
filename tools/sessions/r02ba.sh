#!/bin/bash
O=gpurun_out/r02ba; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "base|X=1" "noprep|SMTTS_ATTN_PREP=0" "single_stream|SMTTS_SINGLE_STREAM=1" "ksplit_enc1|SMTTS_KSPLIT_ENC=1" "shallow|SMTTS_GEMM_DEEP=0" "attn_valu|SMTTS_X=1"; do
  IFS='|' read -r tag envs <<< "$v"
  env TAG=$tag $envs timeout 200 python tools/debug_precision_default.py 2>&1 | grep "repeat" >> $O/out.txt
done
