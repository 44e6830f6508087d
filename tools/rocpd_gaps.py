#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 (rocpd SQLite) kernel trace, for kernels whose name matches a pattern:
    python tools/rocpd_gaps.py x_results.db 'sp_|spl_'
prints, per (previous kernel -> next kernel) pair, the number of transitions and the mean / max gap (next.start - prev.end)."""
import re
import sqlite3
import sys


def main(path, pat):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
    s, e = ('start', 'end') if 'start' in cols else ('start_ts', 'end_ts')
    rows = db.execute("select name, %s, %s from kernels order by %s" % (s, e, s)).fetchall()
    rx = re.compile(pat)
    pairs = {}
    prev = None
    for name, t0, t1 in rows:
        short = re.sub(r'\(anonymous namespace\)::', '', name).split('(')[0][:40]
        if prev is not None and rx.search(prev[0]) and rx.search(short):
            pairs.setdefault((prev[0], short), []).append((t0 - prev[1]) / 1e3)
        prev = (short, t1)
    print("# gaps between consecutive kernels matching %r in %s (microseconds)" % (pat, path))
    for (a, b), g in sorted(pairs.items(), key=lambda kv: -len(kv[1])):
        g.sort()
        print("%-42s -> %-42s n %5d  mean %7.2f  median %7.2f  max %8.2f" % (a, b, len(g), sum(g) / len(g), g[len(g) // 2], g[-1]))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '.')
