#!/bin/bash
O=gpurun_out/r02f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
L=smalltts_amd/libsmalltts_hip
R=3 timeout 1200 bash tools/ab_r02.sh $O "nwv32_4|X=1|$L.so" "nwv32_8|X=1|${L}_g.so" "nwv64_4_minw3|X=1|${L}_f.so"
timeout 2400 python -m pytest tests -q -m gpu > $O/all.txt 2>&1; echo "rc=$?" >> $O/all.txt
