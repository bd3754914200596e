#!/bin/bash
# round 5: degree-3 packed GELU inside the streamed FFN kernels' micro-tasks (A/B vs degree 5 everywhere), interleaved
T=r05j; O=gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_codec_gpu.py tests/test_fullsize_gpu.py -q -x > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
tail -3 $O/tests.txt
D5=$PWD/smalltts_amd/libsmalltts_hip_d5.so
for rep in 1 2 3; do
for v in "deg3:A=1" "deg5:SMTTS_LIB=$D5"; do
  tag=${v%%:*}; envs=${v#*:}
  echo "== $tag" >> $O/ab.txt
  env $envs python tools/phase_breakdown.py --reps 4 2>/dev/null | grep -E "chain|block_wave|ffn_stream|total kernel" >> $O/ab.txt
done; done
cat $O/ab.txt
