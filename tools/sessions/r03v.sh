#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03v.txt; : > $O
timeout 300 python tools/x2_ladder.py >> $O 2>&1
SMTTS_X2_MINK=512 SMTTS_X2_MAXK=512 timeout 300 python tools/x2_ladder.py 2>&1 | grep f16x2 >> $O
SMTTS_X2_MINK=1024 SMTTS_X2_MAXK=1024 timeout 300 python tools/x2_ladder.py 2>&1 | grep f16x2 >> $O
SMTTS_X2_MINK=512 SMTTS_X2_MAXK=4096 timeout 300 python tools/x2_ladder.py 2>&1 | grep f16x2 >> $O
SMTTS_X2_MINK=2048 SMTTS_X2_MAXK=4096 timeout 300 python tools/x2_ladder.py 2>&1 | grep f16x2 >> $O
for i in 1 2 3; do for p in f16 f16,codec_conv=f16x2; do
  printf "%-28s " $p >> $O; python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-roofline --precision $p 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms in flight,", d.get("sequential_ms_per_step"), "one at a time")' >> $O
done; done
timeout 300 python tools/phase_breakdown.py --reps 4 --precision f16,codec_conv=f16x2 2>/dev/null | grep "dec.s[34]\|s4,store\|to_split\|total kernel" >> $O
timeout 900 python -m pytest tests/test_precision_gpu.py tests/test_codec_gpu.py tests/test_fullsize_gpu.py -q -m gpu 2>&1 | tail -4 >> $O
