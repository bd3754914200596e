#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03at.txt; : > $O
L=$PWD/smalltts_amd
timeout 900 python -m pytest tests/test_codec_gpu.py tests/test_fullsize_gpu.py tests/test_kernels_gpu.py tests/test_dit_gpu.py -q -m gpu -x 2>&1 | tail -3 >> $O
for v in libdbg_vv1shallow libsmalltts_hip; do echo "== $v" >> $O; SMTTS_LIB=$L/$v.so timeout 300 python tools/phase_breakdown.py --reps 4 2>/dev/null | grep "gemm<128x128x64\|total kernel" >> $O; done
bash tools/ab_envs.sh 3 "SMTTS_LIB=$L/libdbg_vv1shallow.so" "SMTTS_LIB=$L/libsmalltts_hip.so" >> $O 2>&1
