#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_collect.py tests/test_gpu_sparse.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/sparse_all.log
python bench.py --episodes 64 --agents 1000 --taps 3 --no-cpu-baseline --no-roofline --steps 100 --warmup 10 2>/dev/null | cut -c1-330 >> gpurun_out/sparse_all.log
cat gpurun_out/sparse_all.log
