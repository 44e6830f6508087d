#!/bin/bash
# first GPU pass of round 3: the one-shot exchange, the data-parallel paths, bench on disc vs lattice resets
mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests/test_gpu_p2p.py -x -q 2>&1 | tail -30 > gpurun_out/r3a/p2p.txt
timeout 600 python -m pytest tests/test_gpu_nccl.py -x -q 2>&1 | tail -30 > gpurun_out/r3a/nccl.txt
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -k "bench" 2>&1 | tail -30 > gpurun_out/r3a/bench_tests.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3a/bench_auto_20.json 2> gpurun_out/r3a/bench_auto_20.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --init grid > gpurun_out/r3a/bench_grid_20.json 2> gpurun_out/r3a/bench_grid_20.err
timeout 300 python bench.py --no-cpu-baseline --no-roofline > gpurun_out/r3a/bench_auto_1000.json 2> gpurun_out/r3a/bench_auto_1000.err
timeout 300 python bench.py --dagger --steps 500 --warmup 20 > gpurun_out/r3a/dagger_1.json 2> gpurun_out/r3a/dagger_1.err
MGP_DIST_BACKEND=gloo timeout 300 python bench.py --dagger --gpus 2 --steps 500 --warmup 20 --episodes 128 > gpurun_out/r3a/dagger_2_shared.json 2> gpurun_out/r3a/dagger_2_shared.err
cat gpurun_out/r3a/p2p.txt gpurun_out/r3a/nccl.txt gpurun_out/r3a/bench_tests.txt
