#!/bin/bash
O=gpurun_out/r02ab; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1800 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
timeout 1500 bash tools/profile_round.sh r02ab > $O/profile.log 2>&1
timeout 900 python bench.py > $O/bench_after_profile.json 2> $O/bench2.err
timeout 300 python tools/phase_breakdown.py --reps 4 > $O/phases.txt 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
