#!/bin/bash
# round 5: degree-3 exponent polynomial in the packed-fp16 GELU (A/B against degree 5) + stage chain; codec / full-size tests
T=r05f; O=gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_codec_gpu.py tests/test_fullsize_gpu.py tests/test_precision_gpu.py tests/test_kernels_gpu.py -q -x > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
tail -4 $O/tests.txt
D5=$PWD/smalltts_amd/libsmalltts_hip_d5.so
for rep in 1 2 3; do
for v in "deg3:A=1" "deg5:SMTTS_LIB=$D5"; do
  tag=${v%%:*}; envs=${v#*:}
  printf "%s  " $tag >> $O/ab.txt
  env $envs python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms in flight,", d.get("sequential_ms_per_step"), "one at a time")' >> $O/ab.txt
done; done
for v in "deg3:A=1" "deg5:SMTTS_LIB=$D5"; do
  tag=${v%%:*}; envs=${v#*:}
  echo "== $tag" >> $O/ab.txt
  env $envs python tools/phase_breakdown.py --reps 4 2>/dev/null | grep -E "chain|block_wave|ffn_stream|total kernel" >> $O/ab.txt
done
cat $O/ab.txt
