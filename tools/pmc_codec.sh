#!/bin/bash
# SQ counters of the codec decode kernels (own passes, kernel trace only):  bash tools/pmc_codec.sh <tag> [kernel substring]
TAG=${1:-pmc}; PAT=${2:-codec_ffn}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
           "SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -d gpurun_out/${TAG}_p$i -o c --output-format csv -- python tools/codec_one.py 3 ${PREC:-f16} > /dev/null 2> gpurun_out/${TAG}_p$i.err
  f=$(ls gpurun_out/${TAG}_p$i/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python tools/pmc_kernels.py $f "$PAT"
done
