#!/bin/bash
# round 5: gemm4 (256x256 phase-split GEMM) first light: correctness vs fp64 / gemm3, then the codec shapes
T=r05b; O=gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python tools/gemm4_check.py check > $O/check.txt 2>&1
timeout 900 python tools/gemm4_check.py bench > $O/bench.txt 2>&1
cat $O/check.txt $O/bench.txt
