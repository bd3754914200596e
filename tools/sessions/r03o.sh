#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03o.txt; : > $O
L=$PWD/smalltts_amd
bash tools/ab_envs.sh 3 "SMTTS_LIB=$L/libissue0.so" "SMTTS_LIB=$L/libsmalltts_hip.so" >> $O 2>&1
echo "### sweep, DMA issue before the fragment reads (round 2 order)" >> $O
SMTTS_LIB=$L/libissue0.so timeout 600 python tools/gemm_codec_sweep.py 2>&1 | grep "cfg -1\|cfg  1" >> $O
echo "### sweep, DMA issue after the fragment reads" >> $O
timeout 600 python tools/gemm_codec_sweep.py 2>&1 | grep "cfg -1\|cfg  1" >> $O
SMTTS_LIB=$L/libtimeline.so timeout 600 python tools/gemm3_timeline.py 2>&1 | grep -v amdgpu >> $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_dit_gpu.py -x -q 2>&1 | tail -3 >> $O
