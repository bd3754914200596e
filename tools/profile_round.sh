#!/bin/bash
# Run on the GPU box:  bash tools/profile_round.sh <tag>     -> gpurun_out/<tag>_*  (copy summaries into profiles/)
# rocprofv3 kernel trace + stats of the bench command, then HBM traffic counters in their OWN passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with sys/hip traces), then MFMA-busy.
# Every summary is stamped with the hash of the kernel sources it was measured on (bench.py refuses stale *_latest files).
set -u
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
SHA=$(python -c 'import bench; print(bench.kernel_source_hash())')
CMD="python bench.py --steps 5 --warmup 2 --min-seconds 0 --no-cpu-baseline"
$CMD > gpurun_out/${TAG}_plain_bench.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_trace -o t -- $CMD > gpurun_out/${TAG}_trace_bench.json 2> gpurun_out/${TAG}_trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/${TAG}_fetch -o f --output-format csv -- $CMD --no-roofline > /dev/null 2> gpurun_out/${TAG}_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/${TAG}_write -o w --output-format csv -- $CMD --no-roofline > /dev/null 2> gpurun_out/${TAG}_write.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES -d gpurun_out/${TAG}_mfma -o m --output-format csv -- $CMD --no-roofline > gpurun_out/${TAG}_mfma_bench.json 2> gpurun_out/${TAG}_mfma.err
python tools/rocpd_summary.py $(ls gpurun_out/${TAG}_trace/*.db gpurun_out/${TAG}_trace/*/*.db 2>/dev/null | head -1) gpurun_out/${TAG}_kernel_stats.csv
echo "{\"kernel_src_sha\": \"$SHA\", \"tag\": \"$TAG\", \"cmd\": \"$CMD\"}" > gpurun_out/${TAG}_kernel_stats.meta.json
python tools/pmc_traffic.py $(find gpurun_out/${TAG}_fetch -name '*counter_collection.csv' | head -1) $(find gpurun_out/${TAG}_write -name '*counter_collection.csv' | head -1) gpurun_out/${TAG}_traffic.json $SHA $TAG
# per-batch time of the UN-profiled run (profiled passes run at other clocks and with the counters' overhead)
MS=$(python -c "import json;print(json.loads(open('gpurun_out/${TAG}_plain_bench.json').read().strip().splitlines()[-1])['ms_per_step'])")
# batches per profiled run: warm-up 2 + timed 5 + sequential leg (2 + 5) = 14
python tools/counters_vs_peak.py gpurun_out/${TAG}_traffic.json $(find gpurun_out/${TAG}_mfma -name '*counter_collection.csv' | head -1) 14 $MS gpurun_out/${TAG}_counters_vs_peak.json
# copy into profiles/:  *_kernel_stats.csv + .meta.json (+ as kernel_stats_latest.*), *_traffic.json (+ as traffic_latest.json), *_counters_vs_peak.json, bench JSONs
