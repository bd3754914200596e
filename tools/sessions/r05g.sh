#!/bin/bash
# round 5: gemm3 64x64 with two wave groups sharing each k-tile's k16 steps (KW = 2): tests, kernel A/B, whole batch A/B
T=r05g; O=gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_dit_gpu.py tests/test_precision_gpu.py -q -x > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
tail -4 $O/tests.txt
python - > $O/gemm.txt 2>&1 <<'PY'
import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from smalltts_amd.engine import HipEngine
for kw in ("2", "1"):
    os.environ["SMTTS_GEMM_KW"] = kw
    eng = HipEngine(0, "f16")
    for name, M, N, K, epi in [("dit.out", 600, 960, 960, 3), ("dit.ff2", 600, 960, 2432, 3), ("enc.out", 120, 512, 512, 3), ("dit.qkvg64", 600, 3840, 960, 0)]:
        us = C.c_float()
        rc = eng.lib.smtts_bench_gemm(eng.h, M, N, K, epi, 2, 2, 50, 3, C.byref(us))
        print(f"KW={kw} {name:10s} {M}x{N}x{K} epi {epi} cfg 64x64: {us.value:7.2f} us  {2.0*M*N*K/us.value/1e6:7.1f} TF/s" if not rc else "error", flush=True)
    eng.close()
PY
cat $O/gemm.txt
for rep in 1 2 3; do
for v in "kw2:A=1" "kw1:SMTTS_GEMM_KW=1"; do
  tag=${v%%:*}; envs=${v#*:}
  printf "%s  " $tag >> $O/ab.txt
  env $envs python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms in flight,", d.get("sequential_ms_per_step"), "one at a time")' >> $O/ab.txt
done; done
for v in "kw2:A=1" "kw1:SMTTS_GEMM_KW=1"; do
  tag=${v%%:*}; envs=${v#*:}
  echo "== $tag" >> $O/ab.txt
  env $envs python tools/phase_breakdown.py --reps 4 2>/dev/null | grep -E "64x64|total kernel|^\[dit|^\[enc" >> $O/ab.txt
done
cat $O/ab.txt
