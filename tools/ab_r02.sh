#!/bin/bash
# A/B of builds / env settings on ONE box: bash tools/ab_r02.sh <outdir> "<label>|<env assignments>|<lib>" ...
# each variant: whole-batch ms (in flight / sequential), interleaved R times; then phase breakdown once per variant
OUT=$1; shift
R=${R:-3}
mkdir -p $OUT
for i in $(seq $R); do for v in "$@"; do
  IFS='|' read -r label envs lib <<< "$v"
  printf "%-28s " "$label" >> $OUT/ab.txt
  env $envs SMTTS_LIB=$(realpath ${lib:-smalltts_amd/libsmalltts_hip.so}) python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms in flight,", d.get("sequential_ms_per_step"), "one at a time")' >> $OUT/ab.txt
done; done
for v in "$@"; do
  IFS='|' read -r label envs lib <<< "$v"
  env $envs SMTTS_LIB=$(realpath ${lib:-smalltts_amd/libsmalltts_hip.so}) python tools/phase_breakdown.py --reps 4 > "$OUT/phases_${label// /_}.txt" 2>/dev/null
done
