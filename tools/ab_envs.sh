#!/bin/bash
# A/B several whole environment settings on ONE box, interleaved:  bash tools/ab_envs.sh reps "A=1 B=2" "A=3 B=4" ...
R=$1; shift
for i in $(seq $R); do for cfg in "$@"; do
  printf "%-60s " "$cfg"; env $cfg python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms in flight,", d.get("sequential_ms_per_step"), "one at a time")'
done; done
