#!/bin/bash
# round 5: throughput-mode knobs re-swept on the tree with the stage chain: which persistent kernels take the 192-CU grid cap, cap size, batches in flight
T=r05k; O=gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { printf "%-34s " "$1" >> $O/sweep.txt; shift; env "$@" python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-roofline --no-sequential $EXTRA 2>/dev/null | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms in flight  (min", d["ms_per_step_min"], "max", str(d["ms_per_step_max"]) + ")")' >> $O/sweep.txt; }
for rep in 1 2; do
EXTRA=""
run "default (mask 7, 192 CUs, 3)" A=1
run "mask 5 (wave/chain kernels uncapped)" SMTTS_PERSIST_MASK=5
run "mask 3 (upsample uncapped)" SMTTS_PERSIST_MASK=3
run "mask 1 (only streamed FFN capped)" SMTTS_PERSIST_MASK=1
run "cap 176" SMTTS_PERSIST_CUS=176
run "cap 208" SMTTS_PERSIST_CUS=208
run "cap 224" SMTTS_PERSIST_CUS=224
EXTRA="--in-flight 2"; run "2 in flight" A=1
EXTRA="--in-flight 4"; run "4 in flight" A=1
EXTRA="--in-flight 6"; run "6 in flight" A=1
done
cat $O/sweep.txt
