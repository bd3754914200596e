#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03w.txt; : > $O
L=$PWD/smalltts_amd
for v in libsmalltts_hip libdbg_vnw4 libdbg_vnw4p10 libdbg_vnw4p20 libdbg_vnw8p20; do
  echo "== $v" >> $O
  SMTTS_LIB=$L/$v.so timeout 300 python tools/phase_breakdown.py --reps 4 2>/dev/null | grep "codec_ffn_stream\|total kernel" >> $O
done
bash tools/ab_envs.sh 3 "SMTTS_LIB=$L/libsmalltts_hip.so" "SMTTS_LIB=$L/libdbg_vnw4p20.so" "SMTTS_LIB=$L/libdbg_vnw4p10.so" >> $O 2>&1
