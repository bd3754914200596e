#!/bin/bash
O=gpurun_out/r02bc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "base|X=1" "stage16_off|SMTTS_GEMM_STAGE16=0" "noprep|SMTTS_ATTN_PREP=0" "noprep_again|SMTTS_ATTN_PREP=0" "single|SMTTS_SINGLE_STREAM=1" "stage16_off_shallow|SMTTS_GEMM_STAGE16=0 SMTTS_GEMM_DEEP=0"; do
  IFS='|' read -r tag envs <<< "$v"
  echo "== $tag: $(env $envs timeout 300 python tools/debug_precision_default.py 2>&1 | grep -c 'ref_seq\|phoneme_mem') differing repeats of 24 (ref_seq / phoneme_mem lines)" >> $O/out.txt
done
