#!/usr/bin/env python
"""Where does the host block?  bench.py's flow (pin, model, trainer, fields, 5 warm-up + N steps as one fresh process) with a
watchdog thread that, when the main thread has not finished a step for > 4 ms, makes it dump its C stack (stallwatch.so).
    python tools/diag/stall_hunt.py [steps]      env: HUNT_PIN=0 (no NUMA pinning), HUNT_SLEEP=<s> (idle before stepping)"""
import contextlib, ctypes, io, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from pde_surrogate_amd import parallel
from pde_surrogate_amd.models.codec import DenseED
from pde_surrogate_amd.train import MixedResidualTrainer
from pde_surrogate_amd.utils.data import grf_kle_fields

N = int(sys.argv[1]) if len(sys.argv) > 1 else 45
dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
libc = ctypes.CDLL(None, use_errno=True)
cpu0 = libc.sched_getcpu()


def vmstat():
    out = {}
    with open('/proc/vmstat') as f:
        for line in f:
            k, v = line.split()
            if k.startswith('numa_') or k in ('pgmigrate_success', 'thp_collapse_alloc', 'compact_stall'):
                out[k] = int(v)
    return out


pin = None
if os.environ.get('HUNT_PIN', '1') == '1':
    pin = parallel.pin_rank_to_gpu_numa(dev, 0, 1)
    if os.environ.get('HUNT_MEMPOL', '0') == '1' and pin.get('pinned'):
        node = pin['numa_node']
        nn = len([d for d in os.listdir('/sys/devices/system/node') if d.startswith('node') and d[4:].isdigit()])
        mask = ctypes.c_ulong(1 << node)
        others = ctypes.c_ulong(((1 << nn) - 1) & ~(1 << node))
        t0 = time.perf_counter()
        rc1 = libc.syscall(238, 2, ctypes.byref(mask), 65)                       # set_mempolicy(MPOL_BIND, {node})
        e1 = ctypes.get_errno()
        rc2 = libc.syscall(256, 0, 65, ctypes.byref(others), ctypes.byref(mask))   # migrate_pages(self, others -> node)
        e2 = ctypes.get_errno()
        print(f'mempolicy: set_mempolicy rc {rc1} errno {e1}, migrate_pages rc {rc2} errno {e2}, {1e3 * (time.perf_counter() - t0):.1f} ms, nodes {nn}', flush=True)
torch.manual_seed(1)
with contextlib.redirect_stdout(io.StringIO()):
    model = DenseED(1, 3, 64, [6, 8, 6], growth_rate=16, init_features=48)
tr = MixedResidualTrainer(model, 32, 64, lr=1e-3, weight_bound=10.0, device=dev)
data = torch.from_numpy(grf_kle_fields(4096, cache_dir='/tmp')).to(dev)
perm = torch.randperm(4096, generator=torch.Generator(device='cpu').manual_seed(1)).to(dev)
sw = ctypes.CDLL(os.path.join(ROOT, 'tools', 'diag', 'stallwatch.so'))
sw.sw_install()
hb, stop, pokes = [0], [False], []


def watch():
    last, t_last, poked = hb[0], time.perf_counter(), False
    while not stop[0]:
        time.sleep(0.0005)
        now = time.perf_counter()
        if hb[0] != last:
            last, t_last, poked = hb[0], now, False
        elif now - t_last > 0.004 and not poked and 0 < hb[0] < 38:
            sw.sw_poke()
            pokes.append((hb[0], round((now - t_last) * 1e3, 2)))
            poked = True


def find_bdf():
    import glob
    for d in glob.glob('/sys/class/drm/card*/device'):
        if os.path.exists(os.path.join(d, 'pp_dpm_sclk')):
            return os.path.basename(os.path.realpath(d))
    return None


link_log = []


def link_watch():
    bdf = find_bdf()
    base = f'/sys/bus/pci/devices/{bdf}' if bdf else None
    # the GPU's own link and the upstream port's
    paths = [os.path.join(base, 'current_link_speed'), os.path.join(base, 'current_link_width')] if base else []
    up = os.path.realpath(os.path.join(base, '..')) if base else None
    if up and os.path.exists(os.path.join(up, 'current_link_speed')):
        paths += [os.path.join(up, 'current_link_speed')]
    last = None
    while not stop[0]:
        t0 = time.perf_counter()
        try:
            cur = tuple(open(p_).read().strip() for p_ in paths)
        except OSError as e:
            cur = ('err', str(e))
        t1 = time.perf_counter()
        if cur != last or t1 - t0 > 0.003:
            link_log.append((round(t1 - t_start, 4), hb[0], cur, round((t1 - t0) * 1e3, 2)))
            last = cur
        time.sleep(0.001)


t_start = time.perf_counter()
th = threading.Thread(target=watch, daemon=True)
th2 = threading.Thread(target=link_watch, daemon=True)
if os.environ.get('HUNT_LINK', '0') == '1':
    th2.start()
time.sleep(float(os.environ.get('HUNT_SLEEP', '0')))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 6)]
host = [0.0] * (N + 6)


MAIN_TID = threading.get_native_id()
sched = [None] * (N + 7)


def schedstat():
    with open(f'/proc/self/task/{MAIN_TID}/schedstat') as f:
        a, b, c = f.read().split()
    return int(a), int(b), int(c), libc.sched_getcpu()


def threads():
    out = {}
    for tid in os.listdir('/proc/self/task'):
        try:
            with open(f'/proc/self/task/{tid}/stat') as f:
                st = f.read()
            comm = st[st.index('(') + 1:st.rindex(')')]
            fl = st[st.rindex(')') + 2:].split()
            out[int(tid)] = (comm, int(fl[11]) + int(fl[12]), int(fl[36]))      # utime + stime (ticks), processor
        except (OSError, ValueError):
            pass
    return out


def cg():
    for p_ in ('/sys/fs/cgroup/cpu.stat', '/sys/fs/cgroup/cpu/cpu.stat'):
        try:
            return dict(l.split() for l in open(p_).read().splitlines())
        except OSError:
            pass
    return {}


def step(i):
    lo = (i * 32) % (4096 - 32 + 1)
    tr.load_batch(data, perm[lo:lo + 32])
    tr.step(None, 1e-3)
    ev[i + 1].record()
    host[i + 1] = time.perf_counter()
    sched[i + 1] = schedstat()
    hb[0] = i + 1


try:
    nb = open('/proc/sys/kernel/numa_balancing').read().strip()
except OSError:
    nb = '?'
v0 = vmstat()
ev[0].record()
host[0] = time.perf_counter()
sched[0] = schedstat()
th0, cg0 = threads(), cg()
for i in range(5):
    step(i)
torch.cuda.synchronize()
th.start()
for i in range(5, N + 5):
    step(i)
torch.cuda.synchronize()
stop[0] = True
th1, cg1 = threads(), cg()
v1 = vmstat()
print('numa_balancing', nb, '| started on cpu', cpu0, '| pin', pin, '| vmstat delta during steps',
      {k: v1[k] - v0[k] for k in v1 if v1[k] != v0.get(k, 0)}, flush=True)
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(N + 5)]
hm = [(host[i + 1] - host[i]) * 1e3 for i in range(N + 5)]
print('host stalls > 3 ms (step, ms):', [(i, round(v, 1)) for i, v in enumerate(hm) if v > 3 and i > 0],
      '| gpu steps > 1.9 ms:', [(i, round(v, 2)) for i, v in enumerate(ms) if v > 1.9 and i > 0], '| pokes', pokes, flush=True)
if link_log:
    print('link log (t, step, state, read ms):', link_log[:30], flush=True)
print('stall wall times:', [(i, round(host[i] - t_start, 4), round(host[i + 1] - t_start, 4)) for i, v in enumerate(hm) if v > 3 and i > 0 and i != 5], flush=True)
for i, v in enumerate(hm):
    if v > 3 and i > 0 and i != 5 and i < 40:
        a, b = sched[i], sched[i + 1]
        print(f'step {i}: host {v:.1f} ms: main thread ran {(b[0] - a[0]) / 1e6:.2f} ms, waited on a runqueue {(b[1] - a[1]) / 1e6:.2f} ms, '
              f'{b[2] - a[2]} timeslices, cpu {a[3]} -> {b[3]}', flush=True)
busy = sorted(((th1[t][1] - th0.get(t, ('', 0, 0))[1], t, th1[t][0], th1[t][2]) for t in th1), reverse=True)[:8]
print('threads by cpu ticks during the steps (ticks, tid, comm, last cpu):', busy, '| main tid', MAIN_TID, '| n threads', len(th1), flush=True)
print('affinity of main now:', len(os.sched_getaffinity(0)), 'cpus, on cpu', libc.sched_getcpu(), '| host_threads', getattr(tr, 'host_threads', None), flush=True)
print('cgroup cpu.stat delta:', {k: int(cg1[k]) - int(cg0.get(k, 0)) for k in cg1 if k in ('nr_throttled', 'throttled_usec', 'nr_periods')}, flush=True)
