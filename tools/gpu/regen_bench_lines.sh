#!/bin/bash
# step 2 of tools/regen_profiles.sh alone: the three bench lines (driver form, default, under rocprofv3) into gpurun_out/final/
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd $R; export TMPDIR=/tmp
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err
python bench.py > $O/bench_final.json 2> $O/bench_final.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
T=$(find $O/trace -name "*results.db" | head -1)
cd $R
python tools/rocpd_stats.py $T > $O/bench_kernel_trace.txt 2>&1
rm -rf $O/trace
