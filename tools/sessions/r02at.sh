#!/bin/bash
O=gpurun_out/r02at; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_precision_gpu.py -x -q -m gpu -s -k "ladder or golden" > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
