#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03al.txt; : > $O
L=$PWD/smalltts_amd
for v in libsmalltts_hip libdbg_vlin; do echo "== $v" >> $O; SMTTS_LIB=$L/$v.so timeout 300 python tools/phase_breakdown.py --reps 4 2>/dev/null | grep "codec_ffn_stream\|total kernel" >> $O; done
