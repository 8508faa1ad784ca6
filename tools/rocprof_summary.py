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


def by_grid(path, pattern, out=sys.stdout):
    """per launch size: rows of the kernels view whose name contains `pattern`, grouped by grid (threads) --
    separates e.g. the B = 32 training launches of the loss kernel from the B = 16384 roofline launches"""
    c = sqlite3.connect(path)
    rows = c.execute("select name, grid_x, workgroup_x, count(*), avg(end - start) / 1e3, min(end - start) / 1e3, "
                     "max(end - start) / 1e3 from kernels where name like ? group by name, grid_x order by grid_x",
                     ('%' + pattern + '%',)).fetchall()
    w = csv.writer(out)
    w.writerow(['kernel', 'grid_x_threads', 'workgroups', 'calls', 'avg_us', 'min_us', 'max_us'])
    for name, gx, wx, n, avg, mn, mx in rows:
        name = name if len(name) < 120 else name[:117] + '...'
        w.writerow([name, gx, gx // max(wx, 1), n, round(avg, 3), round(mn, 3), round(mx, 3)])


def per_launch(path, pattern, grid_x, out=sys.stdout):
    """every dispatch of one kernel at one launch size in time order: start offset and duration (the power
    controller's ramp of a long back-to-back sequence is visible here, not in an average)"""
    c = sqlite3.connect(path)
    rows = c.execute("select start, end - start from kernels where name like ? and grid_x = ? order by start",
                     ('%' + pattern + '%', int(grid_x))).fetchall()
    w = csv.writer(out)
    w.writerow(['launch', 'start_us', 'duration_us'])
    for i, (st, du) in enumerate(rows):
        w.writerow([i, round((st - rows[0][0]) / 1e3, 1), round(du / 1e3, 2)])


if __name__ == '__main__':
    if len(sys.argv) > 4 and sys.argv[2] == '--per-launch':
        per_launch(sys.argv[1], sys.argv[3], sys.argv[4])
    elif len(sys.argv) > 3 and sys.argv[2] == '--by-grid':
        by_grid(sys.argv[1], sys.argv[3])
    else:
        main(sys.argv[1])
