#!/usr/bin/env python3
"""Two ranks on ONE device: why is the update time bimodal (15 vs 37-48 us per update)?
Merges the kernel traces of both rank processes (rocprofv3 --kernel-trace writes one rocpd database per process) and prints, for
the update kernels (train_tile_kernel / train_reduce_kernel), per rank: launches, mean duration, and how the two ranks' kernels
sit relative to each other on the device clock -- whether a rank's kernels run WHILE the other rank's polling kernel is resident
(concurrent queues) or only between them (the device alternates between the two processes' queues).
    python tools/gpu/two_rank_trace.py <dir with *_results.db files> [label]"""
import glob
import os
import re
import sqlite3
import sys


def load(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
    s, e = ('start', 'end') if 'start' in cols else ('start_ts', 'end_ts')
    rows = db.execute("select name, %s, %s from kernels order by %s" % (s, e, s)).fetchall()
    out = []
    for name, t0, t1 in rows:
        m = re.search(r'(train_tile_kernel|train_reduce_kernel|replay_aggregate_kernel|replay_gather\w*kernel)', name)
        if m:
            out.append((m.group(1), t0, t1))
    return out


def main(d, label=''):
    dbs = sorted(glob.glob(os.path.join(d, '**', '*results.db'), recursive=True))
    ranks = [(p, load(p)) for p in dbs]
    ranks = [(p, k) for p, k in ranks if len(k) > 100]
    print("# %s: %d process traces with update kernels" % (label or d, len(ranks)))
    if len(ranks) < 2:
        for p, k in ranks:
            print(p, len(k))
        return
    for p, k in ranks[:2]:
        red = [(t1 - t0) / 1e3 for n, t0, t1 in k if n == 'train_reduce_kernel']
        til = [(t1 - t0) / 1e3 for n, t0, t1 in k if n == 'train_tile_kernel']
        red.sort(); til.sort()
        # update period: start of reduce kernel i+1 minus start of reduce kernel i (inside a graph of 32)
        rs = [t0 for n, t0, t1 in k if n == 'train_reduce_kernel']
        per = sorted((b - a) / 1e3 for a, b in zip(rs[:-1], rs[1:]) if (b - a) < 2e5)
        print("%s\n   reduce kernels %d: median %.1f us, p90 %.1f | tile kernels %d: median %.1f us, p90 %.1f | update period median %.1f us, p10 %.1f, p90 %.1f"
              % (os.path.basename(os.path.dirname(p)) + '/' + os.path.basename(p), len(red), red[len(red) // 2], red[int(0.9 * len(red))],
                 len(til), til[len(til) // 2], til[int(0.9 * len(til))], per[len(per) // 2], per[len(per) // 10], per[int(0.9 * len(per))]))
    # overlap: for every tile kernel of rank 1, is a reduce kernel of rank 0 running at its start?
    a, b_ = ranks[0][1], ranks[1][1]
    red0 = [(t0, t1) for n, t0, t1 in a if n == 'train_reduce_kernel']
    til1 = [(t0, t1) for n, t0, t1 in b_ if n == 'train_tile_kernel']
    import bisect
    starts = [t0 for t0, _ in red0]
    inside = 0
    for t0, t1 in til1:
        i = bisect.bisect_right(starts, t0) - 1
        if i >= 0 and red0[i][1] > t0:
            inside += 1
    print("   tile kernels of rank B that START while a reduce (polling) kernel of rank A is running: %d of %d (%.0f %%)"
          % (inside, len(til1), 100.0 * inside / max(1, len(til1))))
    # any two kernels of different ranks overlapping at all
    ev = sorted([(t0, t1, 0) for _, t0, t1 in a] + [(t0, t1, 1) for _, t0, t1 in b_])
    ov = 0.0; busy = 0.0; last_end = [0, 0]
    for t0, t1, r in ev:
        o = min(t1, last_end[1 - r]) - t0
        if o > 0:
            ov += o
        busy += t1 - t0
        last_end[r] = max(last_end[r], t1)
    print("   time in which kernels of BOTH ranks are resident / summed kernel time: %.1f %%" % (100.0 * ov / max(1.0, busy)))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '')
