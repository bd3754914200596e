#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03t.txt; : > $O
bash tools/ab_envs.sh 3 "SMTTS_GEMM_XCD=0 SMTTS_GEMM_GROUP=1" "SMTTS_GEMM_XCD=1 SMTTS_GEMM_GROUP=1" "SMTTS_GEMM_XCD=1 SMTTS_GEMM_GROUP=4" "SMTTS_GEMM_XCD=1 SMTTS_GEMM_GROUP=8" >> $O 2>&1
for e in "SMTTS_GEMM_XCD=0 SMTTS_GEMM_GROUP=1" "SMTTS_GEMM_XCD=1 SMTTS_GEMM_GROUP=4"; do
  echo "== $e" >> $O
  env $e timeout 300 python tools/phase_breakdown.py --reps 4 2>/dev/null | grep "dec.s[0-4]\|,s3,store\|160x128\|128x128" >> $O
done
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_precision_gpu.py tests/test_kernels_gpu.py tests/test_codec_gpu.py -q -m gpu 2>&1 | tail -4 >> $O
