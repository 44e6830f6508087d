#!/usr/bin/env python3
"""First contact with a multi-GPU node, as ONE record (DESIGN section 6 lists what the first run must check):

    bash tools/gpu/first_multi_gpu.sh            # on a node with N >= 2 MI355X: everything below at 1/2/4/8 ranks, RCCL
    DRY=1 bash tools/gpu/first_multi_gpu.sh      # on a one-GPU box: the same commands with 2 ranks sharing the device (gloo carries
                                                 # the handles / barriers); checks the orchestration, never the throughput

  rollout   python bench.py --gpus {1,2,4,8} --steps 20 --warmup 5      value per N -> scaling efficiency vs N = 1
  dagger    python bench.py --dagger --gpus W with MGP_P2P=1 and MGP_P2P=0:  collection rate, us per update, which exchange ran
            (`exchange`: p2p | rccl | gloo), exchange_mem_kind (2 = uncached device memory mapped over hipIpc), the bring-up stage
            if the one-shot exchange was refused, weights bit-identical across ranks
  rccl@2    python bench.py --dagger --gpus 2 with MGP_P2P=0: the update round captured with the RCCL all-reduce inside, two ranks
  exchange  tests/p2p_worker.py allreduce at W ranks, one per device: us per exchange inside a 32-exchange HIP graph
One JSON object on stdout (and gpurun_out/first_multi_gpu.json).  `fallback` is true -- and the exit status 1 -- if MGP_P2P=1 did
NOT run the one-shot exchange (hipIpcOpenMemHandle across devices refused, self-test failed ...): loud, never silent.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_json(cmd, env=None, timeout=900):
    e = dict(os.environ, PYTHONPATH=ROOT)
    e.update(env or {})
    r = subprocess.run(cmd, cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    if r.returncode != 0 or not lines:
        return {"error": "rc %d" % r.returncode, "stderr_tail": r.stderr[-1500:]}
    return json.loads(lines[-1])


def main():
    import torch
    ndev = torch.cuda.device_count()
    dry = os.environ.get('DRY') == '1' or ndev < 2
    rec = {"devices_visible": ndev, "dry_run_on_one_device": dry, "torch": torch.__version__,
           "device_name": torch.cuda.get_device_name(0) if ndev else None}
    share = {'MGP_DIST_BACKEND': 'gloo', 'MGP_P2P_TIMEOUT_MS': '60000'} if dry else {}     # ranks taking turns on one device wait longer
    worlds = [1, 2] if dry else [n for n in (1, 2, 4, 8) if n <= ndev]
    # ---- rollout scaling (no data-path collective: episodes shard)
    rec["rollout"] = {}
    for n in worlds:
        d = run_json([sys.executable, 'bench.py', '--gpus', str(n), '--steps', '20', '--warmup', '5', '--no-cpu-baseline',
                      '--no-roofline'], env=share)
        rec["rollout"][str(n)] = ({"value": d["value"], "ms_per_step": d["ms_per_step"], "dist": d.get("dist"),
                                   "parity_ok": d.get("parity", {}).get("ok")} if "value" in d else d)
    v1 = rec["rollout"].get("1", {}).get("value")
    rec["rollout_scaling_efficiency"] = {n: (r["value"] / (int(n) * v1) if v1 and "value" in r else None)
                                         for n, r in rec["rollout"].items()}
    # ---- the DAGGER round with the one-shot exchange, and with the torch.distributed collective beside it
    W = worlds[-1]
    rec["dagger"] = {}
    for tag, p2p in (("p2p", "1"), ("collective", "0")):
        if dry and p2p == "0":
            rec["dagger"][tag] = {"skipped": "the torch.distributed leg captures an RCCL all-reduce in the update graph; on one device "
                                             "the ranks talk gloo (host), which cannot be captured -- runs on a multi-GPU node only"}
            continue
        d = run_json([sys.executable, 'bench.py', '--dagger', '--gpus', str(W), '--steps', '200', '--warmup', '20',
                      '--episodes', '128' if dry else '256', '--updates', '1024'], env=dict(share, MGP_P2P=p2p))
        if "updates" in d:
            u = d["updates"]
            rec["dagger"][tag] = {"MGP_P2P": p2p, "world": W, "collection_agent_steps_per_s": d["value"],
                                  "us_per_update": 1e3 * u["ms_per_update"], "samples_per_s": u["samples_per_s"],
                                  "exchange": u["exchange"], "exchange_mem_kind": u["exchange_mem_kind"],
                                  "exchange_bringup": u.get("exchange_bringup"),
                                  "weights_bit_identical_across_ranks": d["weights_bit_identical_across_ranks"],
                                  "dist": d["dist"]}
        else:
            rec["dagger"][tag] = d
    # ---- the captured-graph update round on the RCCL all-reduce at TWO ranks (MGP_P2P=0): the form that has only ever run at world 1
    #      (tests/test_gpu_nccl.py) -- its first contact with a second device must not wait for somebody to think of it
    if not dry:
        d = run_json([sys.executable, 'bench.py', '--dagger', '--gpus', '2', '--steps', '200', '--warmup', '20', '--episodes', '256',
                      '--updates', '1024'], env=dict(share, MGP_P2P='0'))
        if "updates" in d:
            u = d["updates"]
            rec["dagger_rccl_2_ranks"] = {"exchange": u["exchange"], "us_per_update": 1e3 * u["ms_per_update"],
                                          "ran_rccl_inside_the_graph": u["exchange"] == 'rccl',
                                          "weights_bit_identical_across_ranks": d["weights_bit_identical_across_ranks"], "dist": d["dist"]}
        else:
            rec["dagger_rccl_2_ranks"] = d
    else:
        rec["dagger_rccl_2_ranks"] = {"skipped": "needs two devices (an RCCL all-reduce between two ranks on one device is not the case to test)"}
    # ---- the exchange alone, W ranks (one per device where there are W devices)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    try:
        import test_gpu_p2p as tp
        extra = {} if dry else {'MGP_DIST_BACKEND': 'nccl'}
        r = tp.run_ranks('allreduce', world=W, timeout=900, **extra)
        rec["exchange_alone"] = {"world": W, "us_per_exchange_in_graph": r["exchange_us_in_graph"], "mem_kind": r["mem_kind"],
                                 "exchanges_checked_bit_exact": r["exchanges"]}
    except BaseException as e:                                   # an assertion inside a rank: keep the record
        rec["exchange_alone"] = {"world": W, "error": repr(e)[-1500:]}
    got = rec["dagger"].get("p2p", {}).get("exchange")
    rec["fallback"] = (W > 1 and got != 'p2p')
    if rec["fallback"]:
        sys.stderr.write("first_multi_gpu: MGP_P2P=1 did NOT run the one-shot exchange (ran: %r, bring-up: %r)\n"
                         % (got, rec["dagger"].get("p2p", {}).get("exchange_bringup")))
    out = json.dumps(rec)
    print(out)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'first_multi_gpu.json'), 'w') as f:
        f.write(out + '\n')
    sys.exit(1 if rec["fallback"] else 0)


if __name__ == '__main__':
    main()
