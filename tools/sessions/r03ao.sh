#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_codec_gpu.py -q -m gpu -k alternative 2>&1 | tail -15 > gpurun_out/r03ao.txt
