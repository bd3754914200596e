#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03z2.txt; : > $O
for e in "SMTTS_MW_TT512=8 SMTTS_MW_TT1024=8 SMTTS_MW_TT2048=8" "SMTTS_MW_TT512=4 SMTTS_MW_TT1024=4 SMTTS_MW_TT2048=4"; do echo "== $e" >> $O; env $e timeout 300 python tools/phase_breakdown.py --reps 4 2>/dev/null | grep "total kernel\|mixer_wide" >> $O; done
SMTTS_MW_TT512=4 SMTTS_MW_TT1024=4 SMTTS_MW_TT2048=4 timeout 600 python -m pytest tests/test_codec_gpu.py -q -m gpu -x 2>&1 | tail -3 >> $O
bash tools/ab_envs.sh 3 "SMTTS_MIXER_WIDE=0" "SMTTS_MIXER_WIDE=1" >> $O 2>&1
