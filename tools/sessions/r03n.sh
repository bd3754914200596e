#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03n.txt; : > $O
SMTTS_LIB=$PWD/smalltts_amd/libtimeline.so timeout 600 python tools/gemm3_timeline.py 2>&1 | grep -v amdgpu >> $O
for k in 2 3 4 5 6 8; do
  printf "in_flight=%s  " $k >> $O
  timeout 300 python bench.py --steps 48 --warmup 6 --in-flight $k --min-seconds 1.5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms/step (", d["ms_per_step_min"], "..", d["ms_per_step_max"], "),", d["value"], "audio-s/s; sequential", d.get("sequential_ms_per_step"))' >> $O
done
