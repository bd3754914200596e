#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -4 > gpurun_out/r03au.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu >> gpurun_out/r03au.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["sequential_ms_per_step"], d["roofline"]["traffic_source"])' >> gpurun_out/r03au.txt
