#!/bin/bash
mkdir -p gpurun_out/r3b
python - <<'PY' > gpurun_out/r3b/nccl_dbg.txt 2>&1
import os, sys, subprocess
sys.path.insert(0, 'tests')
import test_gpu_nccl as t
for mode in ('side', 'current', 'none'):
    for name, code in (('UPDATE', t.UPDATE), ('ROUND', t.ROUND)):
        r = t._run(code, MGP_WARMUP=mode, MGP_P2P='0')
        print('=====', mode, name, 'rc', r.returncode, r.stdout[-300:])
        if r.returncode != 0:
            err = [l for l in r.stderr.splitlines() if 'frame #' not in l]
            print('\n'.join(err[-40:]))
PY
# latency figure of the exchange: two ranks on the one device
python - <<'PY' > gpurun_out/r3b/p2p_latency.txt 2>&1
import sys
sys.path.insert(0, 'tests')
import test_gpu_p2p as t
for w in (2, 3, 4):
    print(w, t.run_ranks('allreduce', world=w))
PY
cat gpurun_out/r3b/nccl_dbg.txt | tail -80; cat gpurun_out/r3b/p2p_latency.txt
