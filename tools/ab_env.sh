#!/bin/bash
# A/B environment settings on ONE box, interleaved:  bash tools/ab_env.sh VAR "A B [C ...]" [reps] [extra bench args]
V=$1; VALS=$2; R=${3:-3}; shift 3 2>/dev/null
for i in $(seq $R); do for x in $VALS; do
  printf "%s=%s  " $V $x; env $V=$x python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms", d.get("sequential_ms_per_step"))'
done; done
