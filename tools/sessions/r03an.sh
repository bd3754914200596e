#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/r03an_bench_driver_style.json 2> gpurun_out/r03an_bench.err
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03an_smoke.txt 2>&1
