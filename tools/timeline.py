#!/usr/bin/env python
"""Per-step timeline from a rocprofv3 rocpd sqlite database (--kernel-trace): for the LAST complete
training step, list every kernel dispatch in start order with start offset, duration, stream/queue and
the gap to the previous dispatch end on the same queue; then the busy/critical summary.
Usage: timeline.py results.db [n_last_dispatches]"""
import sqlite3
import sys


def main(path, nlast=0, marker='adam', first=False):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if 'kernel_dispatch' in t and 'rocpd_kernel_dispatch' in t]
    # the 'kernels' view joins names; fall back to raw table
    if 'kernels' in tabs:
        rows = c.execute('select name, start, end, queue_id, stream_id from kernels order by start').fetchall()
    else:
        raise SystemExit('no kernels view; tables: %s' % tabs)
    # a step = (after the previous adam launch) .. adam launch; take the last complete one
    # (marker: a substring of the kernel that ENDS a step -- or, with `first`, BEGINS one: the drop-in loop's batch gather)
    idx = [i for i, r in enumerate(rows) if marker in r[0]]
    if first and len(idx) >= 3:
        rows = rows[idx[-3]: idx[-2]]
    elif len(idx) >= 2:
        rows = rows[idx[-2] + 1: idx[-1] + 1]
    t0 = rows[0][1]
    last_end = {}
    busy = []
    for name, s, e, q, st in rows:
        gap = (s - last_end.get(st, s)) / 1e3
        last_end[st] = e
        short = name.split('(')[0][-70:]
        print(f'{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f}us  gap {gap:6.1f}  q{q} s{st}  {short}')
        busy.append((s, e))
    total = (max(e for _, e in busy) - t0) / 1e3
    ssum = sum(e - s for s, e in busy) / 1e3
    # union of busy intervals
    busy.sort()
    u, cs, ce = 0, busy[0][0], busy[0][1]
    for s, e in busy[1:]:
        if s > ce:
            u += ce - cs; cs, ce = s, e
        else:
            ce = max(ce, e)
    u += ce - cs
    print(f'step wall {total:.1f} us, sum of kernel durations {ssum:.1f} us, GPU busy (union) {u / 1e3:.1f} us, '
          f'{len(rows)} dispatches')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 400, sys.argv[3] if len(sys.argv) > 3 else 'adam',
         len(sys.argv) > 4 and sys.argv[4] == 'first')
