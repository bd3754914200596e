#!/bin/bash
# (switches other than the ten of DESIGN.md 6a are read by the LAB library only: make -C smalltts_amd/csrc LAB=1, then
#  SMTTS_LIB=$PWD/smalltts_amd/libsmalltts_hip_lab.so bash tools/ab_env.sh ...)
# A/B environment settings on ONE box, interleaved:  bash tools/ab_env.sh VAR "A B [C ...]" [reps] [extra bench args]
V=$1; VALS=$2; R=${3:-3}; shift 3 2>/dev/null
for i in $(seq $R); do for x in $VALS; do
  printf "%s=%s  " $V $x; env $V=$x python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms", d.get("sequential_ms_per_step"))'
done; done
