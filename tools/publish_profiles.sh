#!/bin/bash
# Run HERE after a `bash tools/profile_round.sh <tag>` session came back:  bash tools/publish_profiles.sh <tag>
# copies the stamped summaries from gpurun_out/ into profiles/ (tracked) under their per-round names and as the *_latest files
# bench.py quotes (only when their kernel-source hash equals the tree's).
set -eu
T=$1; G=gpurun_out; P=profiles
for SUF in "" "_latency"; do
  cp $G/${T}_kernel_stats$SUF.csv $P/; cp $G/${T}_kernel_stats$SUF.meta.json $P/
  cp $G/${T}_kernel_stats$SUF.csv $P/kernel_stats${SUF}_latest.csv; cp $G/${T}_kernel_stats$SUF.meta.json $P/kernel_stats${SUF}_latest.meta.json
  cp $G/${T}_traffic$SUF.json $P/; cp $G/${T}_traffic$SUF.json $P/traffic${SUF}_latest.json
done
cp $G/${T}_counters_vs_peak.json $P/ 2>/dev/null || true
cp $G/${T}_traffic_inflight.json $P/ 2>/dev/null || true
python - <<PY
import json, bench
sha = bench.kernel_source_hash()
for f in ("kernel_stats_latest.meta.json", "kernel_stats_latency_latest.meta.json", "traffic_latest.json", "traffic_latency_latest.json"):
    s = json.load(open("profiles/" + f)).get("kernel_src_sha")
    print(f, "OK" if s == sha else f"STALE ({s} != tree {sha})")
PY
