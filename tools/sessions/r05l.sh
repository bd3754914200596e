#!/bin/bash
# round 5: chain warm-up tiles skip their last block: bit-equality + time
T=r05l; O=gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_codec_gpu.py tests/test_fullsize_gpu.py -q -x > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
tail -3 $O/tests.txt
for t in latency throughput; do python tools/chain_check.py 2 75 $t 2>&1 | grep -E "equal"; done | tee $O/equal.txt
for rep in 1 2 3; do python tools/phase_breakdown.py --reps 4 2>/dev/null | grep -E "chain|total kernel"; done | tee $O/time.txt
