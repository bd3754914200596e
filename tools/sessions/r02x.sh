#!/bin/bash
O=gpurun_out/r02x; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "" _exin _exout _exio; do
  echo "== variant '$v'" >> $O/elim.txt
  SMTTS_LIB=$(realpath smalltts_amd/libsmalltts_hip$v.so) timeout 300 python tools/phase_breakdown.py --reps 3 2>/dev/null | grep -E "codec_ffn_stream" >> $O/elim.txt
done
