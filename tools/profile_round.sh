#!/bin/bash
# Run on the GPU box:  bash tools/profile_round.sh <tag>     -> gpurun_out/<tag>_*  (copy summaries into profiles/)
#
# rocprofv3 summaries that bench.py's `roofline` block quotes.  The block describes the TIMED configuration (throughput tuning when
# batches are in flight), so the stamped *_latest files are measured on exactly those kernels, one batch at a time on one stream:
#   kernel_stats_latest.csv / traffic_latest.json                  <- bench.py --in-flight 1 --tuning throughput
#   kernel_stats_latency_latest.csv / traffic_latency_latest.json  <- bench.py --in-flight 1 --tuning latency   (value_sequential)
# HBM traffic counters run in their OWN passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with sys / hip
# traces).  Whole-chip counters (traffic + SQ_VALU_MFMA_BUSY_CYCLES) over the default in-flight bench -> counters_vs_peak.json.
# Every summary is stamped with the hash of the kernel sources it was measured on (bench.py refuses stale *_latest files).
set -u
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
SHA=$(python -c 'import bench; print(bench.kernel_source_hash())')
BASE="python bench.py --steps 5 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-secondary"
ONE="$BASE --no-roofline --in-flight 1 --no-sequential"
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/${TAG}_plain_bench.json 2>/dev/null   # (the un-profiled time per batch: the driver's own steps / warm-up, >= 2 s timed)
db() { ls gpurun_out/$1/*.db gpurun_out/$1/*/*.db 2>/dev/null | head -1; }
cc() { find gpurun_out/$1 -name '*counter_collection.csv' | head -1; }
for T in throughput latency; do
  SUF=$([ $T = throughput ] && echo "" || echo "_latency")
  CMD="$ONE --tuning $T"
  rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_trace$SUF -o t -- $CMD > gpurun_out/${TAG}_trace${SUF}_bench.json 2> gpurun_out/${TAG}_trace$SUF.err
  python tools/rocpd_summary.py $(db ${TAG}_trace$SUF) gpurun_out/${TAG}_kernel_stats$SUF.csv
  echo "{\"kernel_src_sha\": \"$SHA\", \"tag\": \"$TAG\", \"tuning\": \"$T\", \"cmd\": \"$CMD\"}" > gpurun_out/${TAG}_kernel_stats$SUF.meta.json
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/${TAG}_fetch$SUF -o f --output-format csv -- $CMD > /dev/null 2> gpurun_out/${TAG}_fetch$SUF.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/${TAG}_write$SUF -o w --output-format csv -- $CMD > /dev/null 2> gpurun_out/${TAG}_write$SUF.err
  python tools/pmc_traffic.py $(cc ${TAG}_fetch$SUF) $(cc ${TAG}_write$SUF) gpurun_out/${TAG}_traffic$SUF.json $SHA $TAG
done
# whole chip over the in-flight bench (the headline's own command, profile passes off): L2-fill traffic and matrix-pipe busy per batch
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/${TAG}_fetch_if -o f --output-format csv -- $BASE --no-roofline > /dev/null 2> gpurun_out/${TAG}_fetch_if.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/${TAG}_write_if -o w --output-format csv -- $BASE --no-roofline > /dev/null 2> gpurun_out/${TAG}_write_if.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES -d gpurun_out/${TAG}_mfma -o m --output-format csv -- $BASE --no-roofline > gpurun_out/${TAG}_mfma_bench.json 2> gpurun_out/${TAG}_mfma.err
python tools/pmc_traffic.py $(cc ${TAG}_fetch_if) $(cc ${TAG}_write_if) gpurun_out/${TAG}_traffic_inflight.json $SHA $TAG
# per-batch time of the UN-profiled run (profiled passes run at other clocks and with the counters' overhead)
MS=$(python -c "import json;print(json.loads(open('gpurun_out/${TAG}_plain_bench.json').read().strip().splitlines()[-1])['ms_per_step'])")
# batches per profiled in-flight run: warm-up 2 + timed 5 + sequential leg (2 + 5) = 14
python tools/counters_vs_peak.py gpurun_out/${TAG}_traffic_inflight.json $(cc ${TAG}_mfma) 14 $MS gpurun_out/${TAG}_counters_vs_peak.json
# copy into profiles/:  ${TAG}_kernel_stats{,_latency}.csv + .meta.json (+ as kernel_stats{,_latency}_latest.*), ${TAG}_traffic{,_latency}.json
# (+ as traffic{,_latency}_latest.json), ${TAG}_counters_vs_peak.json, the bench JSONs: tools/publish_profiles.sh does it
