#!/bin/bash
# round 5: full GPU suite + default bench on the tree with split-K 2, lazy unpadded QKVG pack, stage chain, degree-3 packed GELU
T=r05i; O=gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 2400 python -m pytest tests -q -m gpu -x > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
tail -4 $O/tests.txt
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python -c "
import json; r=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], r['sequential_ms_per_step'])"
