#!/bin/bash
# round 5: gemm4 per-phase timeline + schedule variants (no setprio / no stagger)
T=r05c; O=gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
SMTTS_LIB=$PWD/smalltts_amd/libsmalltts_hip_tl.so timeout 600 python tools/gemm4_timeline.py > $O/timeline.txt 2>&1
for v in "" _np _ns; do
  echo "== variant '$v'" >> $O/variants.txt
  SMTTS_LIB=$PWD/smalltts_amd/libsmalltts_hip$v.so timeout 600 python tools/gemm4_check.py bench 2>&1 | grep -E "sq4096|s2.ff2|s1.ff1|up.s3" >> $O/variants.txt
done
cat $O/timeline.txt $O/variants.txt
