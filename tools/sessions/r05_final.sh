#!/bin/bash
# Round-5 record of the final tree:  bash tools/sessions/r05_final.sh <tag>   (one gpurun call; copy gpurun_out/<tag>/* into profiles/)
T=${1:-r05z}; O=gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 > $O/bench_driver_style.json 2> $O/bench_driver_style.err
timeout 1500 bash tools/profile_round.sh $T > $O/profile.log 2>&1
timeout 400 python bench.py --workload teacher128 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_teacher128.json 2>/dev/null
timeout 300 python bench.py --workload clone --steps 100 --no-cpu-baseline > $O/bench_clone.json 2>/dev/null
timeout 300 python bench.py --precision bf16x3 --steps 100 --no-cpu-baseline > $O/bench_bf16x3.json 2>/dev/null
timeout 300 python tools/phase_breakdown.py --reps 4 > $O/phases.txt 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
timeout 2400 python -m pytest tests -q -m gpu > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
timeout 900 python tools/stress_determinism.py 32 2>&1 | grep -v amdgpu > $O/stress.txt
SMTTS_DIST_FORCE=1 SMTTS_DIST_BACKEND=nccl timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_torchrun_nccl.json 2> $O/bench_torchrun_nccl.err
