#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03f.txt; : > $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention or qkvg" 2>&1 | tail -15 >> $O
timeout 1500 python -m pytest tests/test_dit_gpu.py tests/test_precision_gpu.py -x -q 2>&1 | tail -15 >> $O
for e in 1 0; do echo "### SMTTS_ATTN_EPI=$e bench" >> $O; SMTTS_ATTN_EPI=$e timeout 600 python bench.py --steps 20 --warmup 3 2>&1 | tail -1 | cut -c1-900 >> $O; done
