#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03u.txt; : > $O
L=$PWD/smalltts_amd
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_precision_gpu.py tests/test_codec_gpu.py -q -m gpu -k "codec or decode or ladder" 2>&1 | tail -4 >> $O
bash tools/ab_envs.sh 3 "SMTTS_LIB=$L/libdbg_vold.so" "SMTTS_LIB=$L/libdbg_vxo.so" "SMTTS_LIB=$L/libdbg_vxnext.so" "SMTTS_LIB=$L/libsmalltts_hip.so" >> $O 2>&1
for v in libdbg_vold libdbg_vxo libdbg_vxnext libsmalltts_hip; do
  echo "== $v" >> $O
  SMTTS_LIB=$L/$v.so timeout 300 python tools/phase_breakdown.py --reps 4 2>/dev/null | grep "codec_ffn_stream\|total kernel" >> $O
done
