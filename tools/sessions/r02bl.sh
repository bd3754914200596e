#!/bin/bash
O=gpurun_out/r02bl; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for p in 0 1; do echo "SMTTS_ATTN_PREP=$p" >> $O/stress.txt; SMTTS_ATTN_PREP=$p timeout 900 python tools/stress_determinism.py 32 2>&1 | grep -v amdgpu >> $O/stress.txt; done
timeout 1800 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
