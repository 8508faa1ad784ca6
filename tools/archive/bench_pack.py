"""time the per-step weight re-packing launch (pdes_pack_all2) as a whole and per table (default DenseED):
    python tools/archive/bench_pack.py"""
import sys, time
sys.path.insert(0, '/root/repo')
import torch
from pde_surrogate_amd import _lib
from pde_surrogate_amd.models.codec import DenseED

dev = torch.device('cuda:0')
net = DenseED(1, 3, 64, [6, 8, 6]).to(dev).train()
net(torch.randn(32, 1, 64, 64, device=dev))              # flattens, builds the tables
L = _lib.lib()
tabs = dict(direct=(net._pack_table, net._pack_n, net._pack_max), mfma=(net._mpack_table, net._mpack_n, net._mpack_max),
            up=(net._upack_table, net._upack_n, net._upack_max), b3=(net._bpack_table, net._bpack_n, net._bpack_max),
            b3up=(net._bupack_table, net._bupack_n, net._bupack_max))
order = ['direct', 'mfma', 'up', 'b3', 'b3up']


def run(active, reps=200):
    args = []
    mx = 1
    for k in order:
        t, n, m = tabs[k]
        if k in active:
            args += [t.data_ptr(), n]
            mx = max(mx, m)
        else:
            args += [None, 0]
    st = _lib.stream_ptr()
    for _ in range(20):
        _lib.check(L.pdes_pack_all2(*args, mx, st), 'pack')
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        L.pdes_pack_all2(*args, mx, st)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


print(f'all tables: {run(order):.1f} us')
for k in order:
    print(f'  only {k:6s} ({tabs[k][1]:2d} items, max {tabs[k][2]}): {run([k]):.1f} us;   all but it: {run([o for o in order if o != k]):.1f} us')
