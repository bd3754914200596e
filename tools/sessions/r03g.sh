#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03g.txt; : > $O
one() { printf "%-55s " "$*" >> $O; env $1 python bench.py --steps 24 --warmup 4 --min-seconds 1.0 --no-cpu-baseline --no-roofline --precision $2 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms in flight (", d["ms_per_step_min"], "..", d["ms_per_step_max"], "),", d.get("sequential_ms_per_step"), "one at a time")' >> $O 2>&1; }
for i in 1 2 3; do
  one SMTTS_ATTN_IMG=0 f16
  one SMTTS_ATTN_IMG=1 f16
  one SMTTS_ATTN_IMG=1 f16,attn=bf16x3
  one SMTTS_ATTN_EPI=0 f16
done
echo "### determinism stress (new default path)" >> $O
timeout 900 python tools/stress_determinism.py 32 2>&1 | grep -v amdgpu >> $O
echo "### full GPU suite" >> $O
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -25 >> $O
