#!/bin/bash
# per-kernel breakdown of the factored path at 64 x 1000 (cfg-3 shape)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/n1000
python bench.py --episodes 64 --agents 1000 --taps 3 --no-cpu-baseline --no-roofline --no-parity --steps 100 --warmup 10 > gpurun_out/n1000/bench.json 2>gpurun_out/n1000/bench.err
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_n1000 -o n1000 -- python $GRAFT_REPO_ROOT/bench.py --episodes 64 --agents 1000 --taps 3 --no-cpu-baseline --no-roofline --no-parity --steps 100 --warmup 10 > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
T=$(find /tmp/prof_n1000 -name "*results.db" | head -1)
python tools/rocpd_stats.py $T > gpurun_out/n1000/kernel_stats.csv 2>&1
python tools/rocpd_gaps.py $T 'sp_|spl_' > gpurun_out/n1000/gaps.txt 2>&1
cat gpurun_out/n1000/gaps.txt
cat gpurun_out/n1000/bench.json | cut -c1-600
cat gpurun_out/n1000/kernel_stats.csv | cut -c1-200
