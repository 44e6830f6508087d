#!/usr/bin/env python3
"""Dump bench.py's rollout state (after `steps` resident steps) for tools/harness/ro_phase_prof.hip:
    python tools/dump_rollout_state.py /tmp/ro_state.bin 220 && RO_STATE=/tmp/ro_state.bin ./ro_prof 256 100 3 200"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    path, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 220
    ro = bench.Rollout(torch.device('cuda:0'), 256, 100, 3, [32, 32], seed=1000)
    ro.run_resident(steps)
    torch.cuda.synchronize()
    with open(path, 'wb') as f:
        f.write(ro.sim.x.cpu().numpy().tobytes())
        f.write(ro.state.delay_gso.cpu().numpy().tobytes())
        f.write(ro.state.delay_state.cpu().numpy().tobytes())
        for conv in ro.actor.conv_layers:
            f.write(conv.weight.detach().reshape(conv.weight.shape[0], -1).contiguous().cpu().numpy().tobytes())
            f.write(conv.bias.detach().cpu().numpy().tobytes())
    print('wrote', path)


if __name__ == '__main__':
    main()
