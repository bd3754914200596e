#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03e.txt; : > $O
timeout 900 python tools/attn_precision.py 2>&1 | grep -v amdgpu.ids >> $O
echo "### old kernels for reference (SMTTS attn_img off is not switchable by env: skipped)" >> $O
timeout 600 python bench.py --steps 20 --warmup 3 2>&1 | tail -1 >> $O
