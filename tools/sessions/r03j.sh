#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03j.txt; : > $O
timeout 900 python tools/gemm_codec_sweep.py 2>&1 | grep -v amdgpu >> $O
timeout 900 python -m pytest tests/test_parallel_gpu.py tests/test_bench_gpu.py -x -q 2>&1 | tail -8 >> $O
