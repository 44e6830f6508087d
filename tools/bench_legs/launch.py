"""bench.py legs: starting the ranks without a launcher, process-group facts, the one JSON line on the original stdout."""
import json
import os
import sys
import time

import torch


def self_launch(n, script):
    """`python bench.py --gpus N` with no launcher: start N ranks of this very command (one per GPU; RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* as torch.distributed.run would set them), pass rank 0's JSON line through, return the worst exit
    status.  Refuses -- loudly -- to run more RCCL ranks than there are devices."""
    import socket
    import subprocess
    backend = os.environ.get('MGP_DIST_BACKEND') or 'nccl'
    have = torch.cuda.device_count()
    if backend == 'nccl' and n > have:
        sys.stderr.write("bench.py: --gpus %d but only %d device(s) visible: RCCL needs one GPU per rank "
                         "(MGP_DIST_BACKEND=gloo lets ranks share a device, for tests)\n" % (n, have))
        return 2
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for rk in range(n):
        env = dict(os.environ, RANK=str(rk), LOCAL_RANK=str(rk), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, script] + sys.argv[1:], env=env,
                                      stdout=None if rk == 0 else subprocess.DEVNULL))
    worst = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                rc = p.poll()
                if rc is None:
                    continue
                pending.remove(p)
                if rc != 0:
                    worst = worst or rc
                    for q in pending:                            # a dead rank leaves the others in a collective: stop them
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return worst


def check_one_device_per_rank(world):
    if world > 1 and torch.distributed.get_backend() == 'nccl' and world > torch.cuda.device_count():
        raise SystemExit("bench.py: %d RCCL ranks but %d device(s) visible: one GPU per rank" % (world, torch.cuda.device_count()))


def dist_record():
    """What the process group really was (the record shows that RCCL saw N ranks)."""
    d = torch.distributed
    if d.is_available() and d.is_initialized():
        return {"backend": d.get_backend(), "world_size": d.get_world_size(), "devices_visible": torch.cuda.device_count()}
    return {"backend": None, "world_size": 1, "devices_visible": torch.cuda.device_count()}


_JSON_FD = [None]


def emit_json(obj):
    """The result line, on the process's ORIGINAL stdout (see main: fd 1 is pointed at stderr while the bench runs)."""
    line = (json.dumps(obj) + "\n").encode()
    if _JSON_FD[0] is None:
        sys.stdout.write(line.decode()); sys.stdout.flush()
    else:
        sys.stdout.flush()
        os.write(_JSON_FD[0], line)
