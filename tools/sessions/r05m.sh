#!/bin/bash
# round 5: streamed FFN at C = 128 with 64 frames per wave (four waves per workgroup, every weight fragment feeds two MFMAs): bit-equality + A/B
T=r05m; O=gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for t in latency throughput; do python tools/env_ab_decode.py SMTTS_FS_COLS 2 1 2 75 $t 2>&1 | grep -v amdgpu; done | tee $O/equal.txt
python tools/env_ab_decode.py SMTTS_FS_COLS 2 1 3 7 2>&1 | grep -v amdgpu | tee -a $O/equal.txt
timeout 900 python -m pytest tests/test_codec_gpu.py tests/test_fullsize_gpu.py -q -x > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
tail -3 $O/tests.txt
for rep in 1 2 3; do for v in 2 1; do echo "== SMTTS_FS_COLS=$v"; SMTTS_FS_COLS=$v python tools/phase_breakdown.py --reps 4 2>/dev/null | grep -E "ffn_stream|total kernel"; done; done | tee $O/time.txt
for rep in 1 2; do for v in 2 1; do printf "cols=$v  "; SMTTS_FS_COLS=$v python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms in flight,", d.get("sequential_ms_per_step"), "one at a time")'; done; done | tee $O/bench.txt
