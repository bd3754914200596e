#!/bin/bash
# A/B two builds of the library on ONE box (see csrc/Makefile):  bash tools/ab_lib.sh libA.so libB.so [reps]
# prints whole-batch ms (in flight / sequential) and the per-kernel ms of the fused FFN kernels for each
A=$1; B=$2; R=${3:-2}
for i in $(seq $R); do for x in $A $B; do
  printf "%s  " $(basename $x); SMTTS_LIB=$(realpath $x) python bench.py --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
kb={k["name"]:k["ms_per_step"] for k in d.get("kernel_breakdown",[])}
print(d["ms_per_step"], "ms", d.get("sequential_ms_per_step"), " ".join(f"{n}={v:.3f}" for n,v in kb.items() if "ffn" in n or "gelu" in n))'
done; done
