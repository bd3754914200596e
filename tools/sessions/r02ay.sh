#!/bin/bash
O=gpurun_out/r02ay; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=tests/test_precision_gpu.py
for k in "site_ladder" "sampler_vs_reference_golden or site_ladder" "vs_reference_golden and not sampler or site_ladder" "small or site_ladder" "cfgrows or site_ladder" "bench1 or site_ladder"; do
  echo "== -k '$k'" >> $O/out.txt
  timeout 300 python -m pytest $T -x -q -m gpu -s -k "$k" 2>&1 | grep -E "f16 \(default\)|passed|failed" >> $O/out.txt
done
