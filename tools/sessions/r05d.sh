#!/bin/bash
# round 5: gemm4 — is the main loop DMA-latency-bound (bytes in flight) or issue-bound?  timing-only variants
T=r05d; O=gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "" _w4 _w63 _nd; do
  echo "== variant '$v'" >> $O/variants.txt
  SMTTS_LIB=$PWD/smalltts_amd/libsmalltts_hip$v.so timeout 600 python tools/gemm4_check.py bench 2>&1 | grep -E "sq4096|sq8192|s2.ff2|s1.ff1|up.s3" >> $O/variants.txt
done
cat $O/variants.txt
