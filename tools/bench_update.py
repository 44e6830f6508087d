#!/usr/bin/env python3
"""DAGGER update throughput (reference learner/gnn_dagger.py:76-96 at cfg/dagger.cfg: B=20, N=100, K=3): thin wrapper
around `python bench.py --dagger-update` (the measurement, including its CPU-port leg, lives in bench.py)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == '__main__':
    import bench
    print(json.dumps(bench.dagger_update_bench()))
