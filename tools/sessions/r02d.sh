#!/bin/bash
# one gpurun session, every stage under its own timeout; results under gpurun_out/r02d/
O=gpurun_out/r02d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_api_gpu.py tests/test_server_gpu.py tests/test_codec_gpu.py -x -q -m gpu > $O/tests_new.txt 2>&1; echo "rc=$?" >> $O/tests_new.txt
timeout 1500 python -m pytest tests/test_bench_gpu.py -x -q -m gpu > $O/tests_bench.txt 2>&1; echo "rc=$?" >> $O/tests_bench.txt
# staged 16-bit epilogue A/B (same binary, env switch), interleaved
for i in 1 2 3; do for v in 1 0; do
  printf "stage16=%s  " $v >> $O/ab_stage16.txt
  SMTTS_GEMM_STAGE16=$v timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms in flight,", d.get("sequential_ms_per_step"), "one at a time")' >> $O/ab_stage16.txt
done; done
SMTTS_GEMM_STAGE16=1 timeout 300 python tools/phase_breakdown.py --reps 4 > $O/phases_stage16_on.txt 2>/dev/null
SMTTS_GEMM_STAGE16=0 timeout 300 python tools/phase_breakdown.py --reps 4 > $O/phases_stage16_off.txt 2>/dev/null
# SQ counters of the fused codec FFN kernels at f16 (own passes)
PREC=f16 timeout 1200 bash tools/pmc_codec.sh r02d_pmc codec_ffn > $O/pmc_codec_ffn_f16.txt 2>&1
timeout 900 python -m pytest tests/test_precision_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > $O/tests_prec.txt 2>&1; echo "rc=$?" >> $O/tests_prec.txt
