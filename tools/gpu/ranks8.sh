#!/bin/bash
# the driver's multi-rank forms at 8 ranks, all on the one GPU of this box (gloo carries the collectives): self-launched and under torchrun
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/r8
{
echo "== python bench.py --gpus 8 (self-launched)"
( time MGP_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/r8/self.err | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('n_gpus', d['n_gpus'], 'value %.3e' % d['value'], 'ms/step %.5f' % d['ms_per_step'], d['dist'], d['config']['episodes_total'], d['parity']['ok'])" ) 2>&1 | tail -5
tail -3 gpurun_out/r8/self.err
echo "== torchrun --nproc-per-node 8 bench.py --gpus 8"
( time MGP_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/r8/torchrun.err | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('n_gpus', d['n_gpus'], 'value %.3e' % d['value'], 'ms/step %.5f' % d['ms_per_step'], d['dist'], d['config']['episodes_total'], d['parity']['ok'])" ) 2>&1 | tail -5
tail -3 gpurun_out/r8/torchrun.err
echo "== python bench.py --dagger --gpus 4 (self-launched, shared GPU)"
( time MGP_DIST_BACKEND=gloo timeout 900 python bench.py --dagger --gpus 4 --steps 100 --warmup 10 --episodes 64 --updates 256 2>gpurun_out/r8/dagger.err | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('n_gpus', d['n_gpus'], 'value %.3e' % d['value'], d['updates'].get('ms_per_update'), d['updates'].get('weights_identical_across_ranks', d['updates'].get('weights_bit_identical')))" ) 2>&1 | tail -5
tail -3 gpurun_out/r8/dagger.err
} > gpurun_out/r8/log.txt 2>&1
cat gpurun_out/r8/log.txt
