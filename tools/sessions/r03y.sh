#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03y.txt; : > $O
timeout 900 python -m pytest tests/test_codec_gpu.py tests/test_fullsize_gpu.py tests/test_precision_gpu.py -q -m gpu -x 2>&1 | tail -6 >> $O
bash tools/ab_envs.sh 3 "SMTTS_MIXER_WIDE=0" "SMTTS_MIXER_WIDE=1" >> $O 2>&1
for e in 0 1; do echo "== SMTTS_MIXER_WIDE=$e" >> $O; SMTTS_MIXER_WIDE=$e timeout 300 python tools/phase_breakdown.py --reps 4 2>/dev/null | grep "total kernel\|dec.s[012]\|mixer_wide\|dwconv\|rmsnorm\|zero_pad" >> $O; done
for e in 0 1; do echo "== clone SMTTS_MIXER_WIDE=$e" >> $O; SMTTS_MIXER_WIDE=$e timeout 300 python bench.py --workload clone --steps 40 --no-cpu-baseline --no-roofline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("sequential_ms_per_step"))' >> $O; done
