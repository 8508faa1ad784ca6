#!/usr/bin/env python
"""Summarise rocprofv3 --pmc results (rocpd sqlite): per kernel name, mean of each counter.
    pmc_summary.py <db>                              text, every pdes kernel
    pmc_summary.py <db> --json <variant> <counter>   one JSON line: the counter of the variant's kernel (tools/bench_loss.py)
    pmc_summary.py --collect <file of such lines>    the profiles/*_loss_variants_pmc.json object bench.py reads"""
import json
import os
import sqlite3
import sys
from collections import defaultdict


def read(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    view = 'counters_collection' if 'counters_collection' in tabs else None
    if view is None:
        return None, tabs
    cur = c.execute(f'select * from {view} limit 1')
    cols = [d[0] for d in cur.description]
    kn = 'kernel_name' if 'kernel_name' in cols else 'name'
    cn = 'counter_name' if 'counter_name' in cols else 'counter'
    vn = 'value' if 'value' in cols else 'counter_value'
    agg = defaultdict(lambda: defaultdict(list))
    for k, n, v in c.execute(f'select {kn}, {cn}, {vn} from {view}'):
        agg[k][n].append(v)
    return agg, tabs


def main(path):
    agg, tabs = read(path)
    if agg is None:
        print('tables:', tabs)
        return
    for k in agg:
        if 'pdes' not in k:
            continue
        print(k[:110])
        for n, vs in sorted(agg[k].items()):
            print(f'    {n:32s} mean {sum(vs) / len(vs):16.1f}  median {sorted(vs)[len(vs) // 2]:16.1f}  n={len(vs)}')


def one(path, variant, counter):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    agg, _ = read(path)
    want = None
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bench_loss.py')) as f:
        src = f.read()
    ns = {}
    exec(src[src.index('PLANE ='):src.index('# the stand-alone Sobel variants')], ns)           # VARIANTS without importing torch
    want = ns['VARIANTS'][variant][2]
    for k in agg or {}:
        if want in k and counter in agg[k]:
            vs = agg[k][counter]
            print(json.dumps({'variant': variant, 'kernel': want, 'counter': counter, 'mean': sum(vs) / len(vs),
                              'median': sorted(vs)[len(vs) // 2], 'min': min(vs), 'max': max(vs), 'dispatches': len(vs)}))
            return
    print(json.dumps({'variant': variant, 'kernel': want, 'counter': counter, 'error': 'kernel or counter not in the database',
                      'kernels': [k[:80] for k in (agg or {}) if 'pdes' in k]}))


def collect(path):
    """FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch; FETCH_SIZE counts HALF of the bytes of these kernels' 16-byte
    per-lane coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM section): doubled.  traffic = 2 * FETCH + WRITE."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bench_loss.py')).read()
    ns = {}
    exec(src[src.index('PLANE ='):src.index('def make_launch')], ns)
    rows = [json.loads(l) for l in open(path) if l.startswith('{')]
    # the fingerprint of the measured kernels' sources: the SAME function bench.py checks the file with (bench_loss.py; its
    # file list, not a copy of it)
    fp_ns = {'os': os, '__file__': os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bench_loss.py')}
    exec(src[src.index('def source_fingerprint'):src.index("if __name__ == '__main__'")], fp_ns)
    out = {'method': 'rocprofv3 --pmc FETCH_SIZE --kernel-trace and rocprofv3 --pmc WRITE_SIZE --kernel-trace in separate passes, one '
                     'process per variant and counter (python tools/bench_loss.py pmc <variant>: 12 dispatches at B = 16384); KiB per '
                     'dispatch, mean over the dispatches; FETCH_SIZE doubled per the gfx950 correction for 16-byte-per-lane coalesced '
                     'reads (MI355X_MICROARCH.md)', 'batch': 16384, 'source_fingerprint': fp_ns['source_fingerprint'](), 'variants': {}}
    by = defaultdict(dict)
    for r in rows:
        by[r['variant']][r['counter']] = r
    for v, d in by.items():
        per_unit, unit, kernel = ns['VARIANTS'][v]
        units = 16384 * ns['UNITS_PER_SAMPLE'].get(v, 1)
        alg = per_unit * units
        e = {'kernel': kernel, 'algorithmic_bytes_per_launch': alg}
        if 'mean' in d.get('FETCH_SIZE', {}) and 'mean' in d.get('WRITE_SIZE', {}):
            rd, wr = 2.0 * d['FETCH_SIZE']['mean'] * 1024.0, d['WRITE_SIZE']['mean'] * 1024.0
            e.update({'FETCH_SIZE_KiB': d['FETCH_SIZE']['mean'], 'WRITE_SIZE_KiB': d['WRITE_SIZE']['mean'], 'hbm_read_bytes': rd,
                      'hbm_write_bytes': wr, 'hbm_bytes_per_launch': rd + wr, 'traffic_over_algorithmic': (rd + wr) / alg,
                      'dispatches': d['FETCH_SIZE']['dispatches']})
        else:
            e['error'] = {c: d.get(c, {}).get('error', 'missing') for c in ('FETCH_SIZE', 'WRITE_SIZE')}
        out['variants'][v] = e
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    if sys.argv[1] == '--collect':
        collect(sys.argv[2])
    elif len(sys.argv) > 2 and sys.argv[2] == '--json':
        one(sys.argv[1], sys.argv[3], sys.argv[4])
    else:
        main(sys.argv[1])
