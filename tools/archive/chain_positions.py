#!/usr/bin/env python
"""Average duration of the k-th launch of a kernel family inside a training step, from a rocprofv3 rocpd database: the
launches whose name contains `pattern`, in time order, folded modulo `per_step` (the family's launches per step).
Usage: chain_positions.py results.db pattern per_step"""
import sqlite3
import sys


def main(path, pattern, per_step):
    c = sqlite3.connect(path)
    rows = c.execute("select name, grid_x, grid_z, workgroup_x, end - start from kernels where name like ? order by start",
                     ('%' + pattern + '%',)).fetchall()
    rows = rows[len(rows) % per_step:]
    n = len(rows) // per_step
    print(f'# {len(rows)} launches = {n} steps x {per_step}')
    print('position,kernel,workgroups,avg_us,min_us')
    for k in range(per_step):
        grp = rows[k::per_step]
        d = [g[4] / 1e3 for g in grp]
        nm = grp[0][0]
        nm = nm[nm.index('<'):nm.index('(')] if '<' in nm else nm
        print(f'{k},{nm},{grp[0][1] // max(grp[0][3], 1) * max(grp[0][2], 1)},{sum(d) / len(d):.2f},{min(d):.2f}')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]))
