#!/bin/bash
# A/B two environment settings on ONE box, interleaved:  bash tools/ab_env.sh VAR A B [reps]
V=$1; A=$2; B=$3; R=${4:-3}
for i in $(seq $R); do for x in $A $B; do
  printf "%s=%s  " $V $x; env $V=$x python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms")'
done; done
