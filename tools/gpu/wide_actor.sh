cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/wide
{
timeout 1500 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_actor.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -3
for cfg in "256 100 3 128 2" "256 100 3 64 2" "256 100 3 128 1"; do set -- $cfg
for st in 20 100; do
python bench.py --episodes $1 --agents $2 --taps $3 --hidden $4 --layers $5 --no-cpu-baseline --no-roofline --steps $st --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); p = d['parity']
print('$cfg steps $st | value %.3e |' % d['value'], ' '.join('%s %.3e' % (a, b['value']) for a, b in d['paths'].items()), '| parity', p['ok'], {k: '%.2e' % v['max_rel'] for k, v in p['paths'].items()})"
done; done
MGP_ACTOR_WIDE=0 python bench.py --episodes 256 --agents 100 --taps 3 --hidden 128 --layers 2 --no-cpu-baseline --no-roofline --steps 100 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('MGP_ACTOR_WIDE=0 [128,128] steps 100 | value %.3e' % d['value'])"
MGP_ACTOR_WIDE=0 python bench.py --episodes 256 --agents 100 --taps 3 --hidden 64 --layers 2 --no-resident --no-cpu-baseline --no-roofline --steps 100 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('MGP_ACTOR_WIDE=0 [64,64] two-launch steps 100 | value %.3e' % d['value'])"
python bench.py --episodes 256 --agents 100 --taps 3 --hidden 64 --layers 2 --no-resident --no-cpu-baseline --no-roofline --steps 100 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('[64,64] two-launch steps 100 | value %.3e' % d['value'])"
} > gpurun_out/wide/run.log 2>&1
cat gpurun_out/wide/run.log
