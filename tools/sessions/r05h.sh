#!/bin/bash
# round 5: (1) VERDICT r4 item 4 table — the N = 960 projections in latency tuning: split-K 3 (today) / 2 / unsplit + ln_modulate;
# (2) SQ counters of gemm4 vs gemm3 on the codec's two named shapes + 4096^3; (3) HBM traffic of the C = 32 stage: chain vs one launch per block
T=r05h; O=gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
for v in "split3:A=1" "split2:SMTTS_KSPLIT_OUT=2 SMTTS_KSPLIT_FF2=2" "unsplit:SMTTS_KSPLIT_OUT=1 SMTTS_KSPLIT_FF2=1"; do
  tag=${v%%:*}; envs=${v#*:}
  printf "%s  " $tag >> $O/splitk.txt
  env $envs python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms in flight,", d.get("sequential_ms_per_step"), "one at a time")' >> $O/splitk.txt
done; done
for v in "split3:A=1" "split2:SMTTS_KSPLIT_OUT=2 SMTTS_KSPLIT_FF2=2" "unsplit:SMTTS_KSPLIT_OUT=1 SMTTS_KSPLIT_FF2=1"; do
  tag=${v%%:*}; envs=${v#*:}
  echo "== $tag (latency tuning, kernel time per batch by HIP events)" >> $O/splitk.txt
  env $envs python tools/phase_breakdown.py --reps 4 2>/dev/null | grep -E "^\[dit|64x64|splitk_resid|ln_modulate|total kernel" >> $O/splitk.txt
done
cat $O/splitk.txt
# (2) counters
for sh in "24000 2048 512 4" "4800 4096 1024 4" "4096 4096 4096 0"; do
 for cfg in 1 7; do
  i=0
  for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
             "SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $SET -d $O/pmc_g_${cfg}_$i -o c --output-format csv -- python tools/gemm_pmc_one.py $sh $cfg > /dev/null 2> $O/pmc.err
    f=$(find $O/pmc_g_${cfg}_$i -name '*counter_collection.csv' | head -1)
    echo "## shape $sh cfg $cfg set $i" >> $O/gemm_counters.txt
    [ -n "$f" ] && python tools/pmc_kernels.py $f gemm >> $O/gemm_counters.txt
    rm -rf $O/pmc_g_${cfg}_$i
  done
 done
done
tail -80 $O/gemm_counters.txt
# (3) traffic of the C = 32 stage
for v in "chain:A=1" "blocks:SMTTS_STAGE_CHAIN=0"; do
  tag=${v%%:*}; envs=${v#*:}
  for SET in FETCH_SIZE WRITE_SIZE; do
    env $envs timeout 300 rocprofv3 --kernel-trace --pmc $SET -d $O/pmc_c -o c --output-format csv -- python tools/codec_one.py 3 f16 > /dev/null 2>> $O/pmc.err
    f=$(find $O/pmc_c -name '*counter_collection.csv' | head -1)
    echo "## $tag $SET" >> $O/chain_traffic.txt
    [ -n "$f" ] && python tools/pmc_kernels.py $f wave >> $O/chain_traffic.txt
    rm -rf $O/pmc_c
  done
done
cat $O/chain_traffic.txt
