#!/bin/bash
# Run on the GPU box:  bash tools/profile_round.sh <tag>     -> gpurun_out/<tag>_*  (copy summaries into profiles/)
# rocprofv3 kernel trace + stats of the bench command, then HBM traffic counters in their OWN passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with sys/hip traces).
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
CMD="python bench.py --steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_trace -o t -- $CMD > gpurun_out/${TAG}_trace_bench.json 2> gpurun_out/${TAG}_trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/${TAG}_fetch -o f --output-format csv -- $CMD --no-roofline > /dev/null 2> gpurun_out/${TAG}_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/${TAG}_write -o w --output-format csv -- $CMD --no-roofline > /dev/null 2> gpurun_out/${TAG}_write.err
python tools/rocpd_summary.py $(ls gpurun_out/${TAG}_trace/*.db | head -1) gpurun_out/${TAG}_kernel_stats.csv
python tools/pmc_traffic.py gpurun_out/${TAG}_fetch/f_counter_collection.csv gpurun_out/${TAG}_write/w_counter_collection.csv gpurun_out/${TAG}_traffic.json
# copy into profiles/ by hand:  *_kernel_stats.csv (+ as kernel_stats_latest.csv), *_traffic.json (+ as traffic_latest.json), bench JSONs
