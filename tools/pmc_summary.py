#!/usr/bin/env python
"""Summarise rocprofv3 --pmc results (rocpd sqlite): per kernel name, mean of each counter."""
import sqlite3
import sys
from collections import defaultdict


def main(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    view = 'counters_collection' if 'counters_collection' in tabs else None
    if view is None:
        print('tables:', tabs)
        return
    cur = c.execute(f'select * from {view} limit 1')
    cols = [d[0] for d in cur.description]
    kn = 'kernel_name' if 'kernel_name' in cols else 'name'
    cn = 'counter_name' if 'counter_name' in cols else 'counter'
    vn = 'value' if 'value' in cols else 'counter_value'
    agg = defaultdict(lambda: defaultdict(list))
    for k, n, v in c.execute(f'select {kn}, {cn}, {vn} from {view}'):
        agg[k][n].append(v)
    for k in agg:
        if 'pdes' not in k:
            continue
        print(k[:110])
        for n, vs in sorted(agg[k].items()):
            print(f'    {n:32s} mean {sum(vs) / len(vs):16.1f}  median {sorted(vs)[len(vs) // 2]:16.1f}  n={len(vs)}')


if __name__ == '__main__':
    main(sys.argv[1])
