#!/bin/bash
# full-size configs[4] variants
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "variants" -s 2>&1 | tail -30 > gpurun_out/variants.log
cat gpurun_out/variants.log
