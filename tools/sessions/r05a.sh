#!/bin/bash
# round 5, first GPU session: the measurement changes (roofline under the timed tuning, --gpus relaunch), new tests, baseline numbers
T=r05a; O=gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 2400 python -m pytest tests -q -m gpu -x > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
timeout 300 python tools/phase_breakdown.py --reps 4 > $O/phases.txt 2>/dev/null
tail -3 $O/tests.txt; tail -c 1500 $O/bench.json
