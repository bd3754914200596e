#!/bin/bash
# A/B two builds of the library on ONE box (see csrc/Makefile):  bash tools/ab_lib.sh libA.so libB.so [reps] [kernel-name pattern]
# prints whole-batch ms (in flight / sequential) for each, then the per-kernel lines of tools/phase_breakdown.py that match
A=$1; B=$2; R=${3:-2}; PAT=${4:-ffn}
for i in $(seq $R); do for x in $A $B; do
  printf "%s  " $(basename $x); SMTTS_LIB=$(realpath $x) python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms in flight,", d.get("sequential_ms_per_step"), "one at a time")'
done; done
for x in $A $B; do echo "== $(basename $x)"; SMTTS_LIB=$(realpath $x) python tools/phase_breakdown.py --reps 4 2>/dev/null | grep -E "$PAT|total kernel"; done
