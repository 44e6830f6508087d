#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd SQLite) kernel trace: per-kernel calls / total / avg / min / max duration.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [> profiles/rNN_name.txt]
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return name if len(name) <= 110 else name[:107] + '...'


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                      "max(vgpr_count), max(lds_size), max(grid_x*1.0/workgroup_x*grid_y/workgroup_y*grid_z/workgroup_z) "
                      "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("# rocprofv3 --kernel-trace summary of %s" % path)
    print("# durations in microseconds; pct = share of total GPU kernel time")
    print("%-112s %8s %12s %10s %10s %10s %6s %5s %8s %7s" % ('kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us',
                                                            'pct', 'vgpr', 'lds_B', 'wgs'))
    for name, calls, tot, avg, mn, mx, vg, lds, wgs in rows:
        print("%-112s %8d %12.1f %10.2f %10.2f %10.2f %6.2f %5s %8s %7d" % (short(name), calls, tot / 1e3, avg / 1e3, mn / 1e3,
                                                                          mx / 1e3, 100.0 * tot / total, vg, lds, wgs or 0))


if __name__ == '__main__':
    main(sys.argv[1])
