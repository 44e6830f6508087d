#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/final; mkdir -p $O
python bench.py --dagger --episodes 64 --agents 1000 --steps 200 --warmup 10 --updates 64 2> $O/dagger_round_n1000.err | grep "^{" > $O/dagger_round_n1000.json
python bench.py --dagger --episodes 256 --agents 300 --steps 200 --warmup 10 --updates 256 2> $O/dagger_round_n300.err | grep "^{" > $O/dagger_round_n300.json
python bench.py --dagger --steps 100 --warmup 10 --updates 64 2>/dev/null | cut -c1-300
tail -2 $O/dagger_round_n1000.err; cut -c1-1800 $O/dagger_round_n1000.json; cut -c1-400 $O/dagger_round_n300.json
