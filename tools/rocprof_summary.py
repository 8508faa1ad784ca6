#!/usr/bin/env python
"""Dump the per-kernel statistics of a rocprofv3 rocpd sqlite database (the `--kernel-trace --stats`
summary) as CSV: name, calls, total_us, avg_us, pct.  Usage: rocprof_summary.py results.db > out.csv"""
import csv
import sqlite3
import sys


def main(path, out=sys.stdout, limit=40):
    c = sqlite3.connect(path)
    rows = c.execute('select name, total_calls, total_duration, average, percentage from top_kernels').fetchall()
    w = csv.writer(out)
    w.writerow(['kernel', 'calls', 'total_us', 'avg_us', 'pct'])
    for name, calls, tot, avg, pct in rows[:limit]:
        name = name if len(name) < 160 else name[:157] + '...'
        w.writerow([name, calls, round(tot, 3), round(avg, 3), round(pct, 3)])


if __name__ == '__main__':
    main(sys.argv[1])
