#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (rocpd .db) into per-kernel HBM traffic per launch.

gfx950 corrections (MI355X_MICROARCH.md, section HBM): counter values are KiB; FETCH_SIZE reports exactly 1/2
of the bytes of a wide coalesced streaming read -> doubled.  WRITE_SIZE is taken as is (checked here against the
known output size of agg_fwd: 1.84 MB algorithmic vs 1.92 MB counted).

    python tools/pmc_summary.py gpurun_out/pmc/fetch_results.db gpurun_out/pmc/write_results.db [out.json]
"""
import json
import re
import sqlite3
import sys


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? "
                      "group by kernel_name", (counter,)).fetchall()
    out = {}
    for name, n, avg in rows:
        m = re.search(r'(\w+_kernel)', name)
        if m and 'at::native' not in name:
            out[m.group(1)] = (n, avg)
    return out


def main():
    fetch = per_kernel(sys.argv[1], 'FETCH_SIZE')
    write = per_kernel(sys.argv[2], 'WRITE_SIZE')
    res = {}
    print("# HBM traffic per launch from rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE passes)")
    print("# read = 2 x FETCH_SIZE KiB (gfx950 half-count correction), write = WRITE_SIZE KiB")
    print("%-28s %8s %14s %14s %14s" % ('kernel', 'launches', 'read_MB', 'write_MB', 'total_MB'))
    for k in sorted(set(fetch) | set(write)):
        rd = 2.0 * fetch.get(k, (0, 0.0))[1] * 1024.0
        wr = write.get(k, (0, 0.0))[1] * 1024.0
        res[k] = dict(read_bytes=rd, write_bytes=wr, total_bytes=rd + wr, launches=fetch.get(k, (0, 0))[0])
        print("%-28s %8d %14.3f %14.3f %14.3f" % (k, res[k]['launches'], rd / 1e6, wr / 1e6, (rd + wr) / 1e6))
    # optional: a second FETCH/WRITE pair taken with another PROBE_T (argv[5], argv[6], its T in argv[7]) turns the
    # resident kernel's bytes per launch into bytes(T) = fixed + per_step * T (state in/out once, 8 B of reward per step)
    if len(sys.argv) > 7:
        f2 = per_kernel(sys.argv[5], 'FETCH_SIZE'); w2 = per_kernel(sys.argv[6], 'WRITE_SIZE')
        import os as _os
        T1, T2 = int(_os.environ.get('PROBE_T', '1000')), int(sys.argv[7])
        for k in list(res):
            if k.startswith('rollout') and k in f2 and k in w2:
                b1 = res[k]['total_bytes']
                b2 = 2.0 * f2[k][1] * 1024.0 + w2[k][1] * 1024.0
                per_step = (b1 - b2) / float(T1 - T2)
                res[k + '_model'] = dict(bytes_per_step=per_step, fixed_bytes=b1 - per_step * T1, T=[T1, T2], bytes=[b1, b2])
                print("%-28s bytes(T) = %.0f + %.1f * T   (from launches of %d and %d steps: %.3f MB, %.3f MB)" % (
                    k + '_model', b1 - per_step * T1, per_step, T1, T2, b1 / 1e6, b2 / 1e6))
    if len(sys.argv) > 3:
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from multiagent_gnn_policies_amd import build as mgp_build
        shape = [int(v) for v in sys.argv[4].split(',')] if len(sys.argv) > 4 else [256, 100, 3]
        res['_meta'] = {'shape': shape, 'source_hash': mgp_build.source_hash(), 'rollout_steps_per_launch': int(os.environ.get('PROBE_T', '1000')),
                        'factored_steps_per_launch': int(os.environ.get('PROBE_FT', '200')),
                        'note': 'bytes per launch; read = 2 x FETCH_SIZE KiB, write = WRITE_SIZE KiB'}
        with open(sys.argv[3], 'w') as f:
            json.dump(res, f, indent=1)


if __name__ == '__main__':
    main()
