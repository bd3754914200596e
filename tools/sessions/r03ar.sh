#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03ar.txt; : > $O
L=$PWD/smalltts_amd
for v in libsmalltts_hip libdbg_vupop; do echo "== $v" >> $O; SMTTS_LIB=$L/$v.so timeout 300 python tools/phase_breakdown.py --reps 4 2>/dev/null | grep "upsample\|total kernel" >> $O; done
