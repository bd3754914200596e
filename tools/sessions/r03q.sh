#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03q2.txt; : > $O
L=$PWD/smalltts_amd
bash tools/ab_envs.sh 4 "SMTTS_LIB=$L/libprev.so" "SMTTS_LIB=$L/libsmalltts_hip.so" >> $O 2>&1
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_dit_gpu.py tests/test_codec_gpu.py tests/test_fullsize_gpu.py -x -q 2>&1 | tail -3 >> $O
timeout 300 python tools/phase_breakdown.py --reps 4 2>/dev/null | head -14 >> $O
