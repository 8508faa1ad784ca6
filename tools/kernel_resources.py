#!/usr/bin/env python
"""Resource footprint of every kernel of a rocprofv3 rocpd database (what decides whether a workgroup of the main
chain fits on a CU beside the weight-gradient streams' workgroups): registers, LDS, workgroups, average duration.
Usage: kernel_resources.py results.db"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute('pragma table_info(kernels)').fetchall()]
    want = [k for k in ('name', 'grid_x', 'grid_y', 'grid_z', 'workgroup_x', 'lds_size', 'lds_block_size', 'static_lds_size',
                        'dynamic_lds_size', 'vgpr_count', 'arch_vgpr_count', 'accum_vgpr_count', 'sgpr_count', 'scratch_size') if k in cols]
    print('# columns of the kernels view:', ','.join(cols))
    q = 'select ' + ','.join(want) + ', count(*), avg(end - start) / 1e3 from kernels group by ' + ','.join(want) + ' order by sum(end - start) desc'
    print(','.join(want) + ',calls,avg_us')
    for r in c.execute(q).fetchall()[:60]:
        r = list(r)
        r[0] = r[0] if len(r[0]) < 100 else r[0][:97] + '...'
        r[-1] = round(r[-1], 2)
        print(','.join(str(x) for x in r))


if __name__ == '__main__':
    main(sys.argv[1])
