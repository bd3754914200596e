#!/bin/bash
# round 5: stage chain at C = 32 (three blocks per tile in one launch): bit-equality tests, per-kernel times, whole batch A/B
T=r05e; O=gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_codec_gpu.py tests/test_fullsize_gpu.py -q -x > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
tail -5 $O/tests.txt
for rep in 1 2; do
for v in "chain12:A=1" "nochain:SMTTS_STAGE_CHAIN=0" "chain8:SMTTS_LIB=$PWD/smalltts_amd/libsmalltts_hip_c8.so"; do
  tag=${v%%:*}; envs=${v#*:}
  printf "%s  " $tag >> $O/ab.txt
  env $envs python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms in flight,", d.get("sequential_ms_per_step"), "one at a time")' >> $O/ab.txt
done; done
for v in "chain12:A=1" "nochain:SMTTS_STAGE_CHAIN=0" "chain8:SMTTS_LIB=$PWD/smalltts_amd/libsmalltts_hip_c8.so"; do
  tag=${v%%:*}; envs=${v#*:}
  echo "== $tag" >> $O/ab.txt
  env $envs python tools/phase_breakdown.py --reps 4 2>/dev/null | grep -E "dec.s6|chain|block_wave<32>|upsample_wave<128|total kernel" >> $O/ab.txt
done
cat $O/ab.txt
