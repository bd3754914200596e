#!/bin/bash
O=gpurun_out/r02w; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "" _eGELU _eMFMA _eFRAG _enb _egm; do
  echo "== variant '$v'" >> $O/elim.txt
  SMTTS_LIB=$(realpath smalltts_amd/libsmalltts_hip$v.so) timeout 300 python tools/phase_breakdown.py --reps 3 2>/dev/null | grep -E "codec_ffn_stream|codec_block_wave" >> $O/elim.txt
done
