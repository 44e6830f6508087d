#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_collect.py tests/test_gpu_sparse.py -x -q -m gpu -k "${K:-sparse or factored}" 2>&1 | tail -40 > gpurun_out/collect_sparse.log
cat gpurun_out/collect_sparse.log
