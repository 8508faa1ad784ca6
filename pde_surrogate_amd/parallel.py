"""Host-side data-parallel logic of the mixed-residual trainer (one process per GPU).

The reference has no distributed code at all; data parallelism is what this build adds:
  * every rank holds the dataset replica and the same permutation; rank r takes samples
    [step*GB + r*B, step*GB + (r+1)*B) of it (`shard_indices`);
  * BatchNorm statistics stay rank-local (what DistributedDataParallel does);
  * gradients are ONE flat fp32 buffer -> one all-reduce(SUM) per step over RCCL/xGMI (2.96 MB for the
    default net: latency-bound), then Adam with grad_scale = 1/world_size on every rank;
  * parameters start identical (same seed, checked/enforced by `broadcast_parameters`).
Rendezvous, broadcasts and the once-per-epoch scalar means go through `torch.distributed` (backend `nccl` = RCCL on
ROCm; the CPU tests run them over `gloo` with world_size 2).  The PER-STEP gradient exchange does not: `DirectRccl`
holds its own RCCL communicator and enqueues `ncclAllReduce` by pointer on the stream where the data becomes final (ctypes into the librccl.so
torch ships) -- a `torch.distributed.all_reduce` costs the host ~0.3 ms per call (Work objects, stream guards, event
bookkeeping: 1.63 vs 0.97 ms of host time per step measured with two buckets on one rank), the direct call a few
microseconds, and the step of an 8-rank job must not become host-bound.
"""
import ctypes
import os
import socket
import sys

import torch
import torch.distributed as dist


class _NcclUniqueId(ctypes.Structure):
    _fields_ = [('internal', ctypes.c_char * 128)]


class DirectRccl:
    """one RCCL communicator over the ranks of a torch process group (backend nccl), driven through ctypes.
    `all_reduce_sum_(ptr, count, stream)` enqueues an in-place fp32 SUM all-reduce on the given hipStream_t."""
    NCCL_FLOAT32, NCCL_SUM = 7, 0

    def __init__(self, group=None, device=None):
        if not (dist.is_available() and dist.is_initialized()) or dist.get_backend(group) != 'nccl':
            raise RuntimeError('DirectRccl needs an initialised torch.distributed process group with backend nccl (RCCL)')
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self._comm = None
        # 1. everything that can fail on ONE rank only (loading the library, ncclGetUniqueId on rank 0) happens before any
        #    collective, and its outcome is exchanged together with the id: either every rank goes on to
        #    ncclCommInitRank or every rank raises -- no rank is left waiting in a collective the others never enter
        uid, err = _NcclUniqueId(), None
        try:
            path = os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')
            L = ctypes.CDLL(path)                            # the library torch itself loaded: the same RCCL, the same HIP runtime
            L.ncclGetUniqueId.argtypes = [ctypes.POINTER(_NcclUniqueId)]
            L.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _NcclUniqueId, ctypes.c_int]
            L.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_void_p, ctypes.c_void_p]
            L.ncclCommDestroy.argtypes = [ctypes.c_void_p]
            L.ncclGetErrorString.argtypes = [ctypes.c_int]
            L.ncclGetErrorString.restype = ctypes.c_char_p
            self._L = L
            if self.rank == 0:
                self._check(L.ncclGetUniqueId(ctypes.byref(uid)), 'ncclGetUniqueId')
        except Exception as e:                               # noqa: BLE001
            err = e
        status = [None] * self.world
        dist.all_gather_object(status, (None if err is None else f'{type(err).__name__}: {err}',
                                        bytes(bytearray(uid)) if self.rank == 0 and err is None else None), group=group)
        failed = [(r, s[0]) for r, s in enumerate(status) if s[0] is not None]
        if failed:
            raise RuntimeError('DirectRccl: local setup failed on rank(s) ' + ', '.join(f'{r} ({m})' for r, m in failed))
        ctypes.memmove(ctypes.byref(uid), status[0][1], 128)
        # 2. the collective part
        comm = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            self._check(L.ncclCommInitRank(ctypes.byref(comm), self.world, uid, self.rank), 'ncclCommInitRank')
        self._comm = comm

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f'{what} failed: {self._L.ncclGetErrorString(rc).decode()} ({rc})')

    def all_reduce_sum_(self, ptr, count, stream_ptr):
        rc = self._L.ncclAllReduce(ptr, ptr, count, self.NCCL_FLOAT32, self.NCCL_SUM, self._comm, stream_ptr)
        if rc != 0:
            self._check(rc, 'ncclAllReduce')

    def close(self):
        """destroy the communicator (idempotent; MixedResidualTrainer.close / __del__ call it)"""
        if getattr(self, '_comm', None):
            self._L.ncclCommDestroy(self._comm)
            self._comm = None

    def __del__(self):
        # garbage collection only.  At interpreter shutdown the communicator is left to the process exit: ncclCommDestroy
        # may block on a peer that has already gone, and a hang inside the C call cannot be caught -- callers that want a
        # clean teardown call close() (before destroy_process_group)
        if sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:                                    # noqa: BLE001
            pass


class CollectiveWatchdog:
    """A collective that never completes (a peer died, a rank took another branch) blocks its stream for ever; the host keeps
    enqueueing until the queue is full and then hangs inside a HIP call no exception can leave.  This daemon thread turns that
    into an error: `arm(what, done)` registers a collective just enqueued with a callable that is True once it has completed
    (an event's `query`); one still outstanding after `timeout` seconds is reported -- rank, host, what, how long -- through
    `on_timeout(message)`, by default a line on stderr and `os._exit(13)` so that the launcher tears the job down.
    PDES_DP_TIMEOUT_S sets the timeout (default 30 s; 0 disables the watchdog)."""

    def __init__(self, rank=0, world=1, timeout=None, on_timeout=None, poll=0.5):
        import collections
        import threading
        self.rank, self.world = rank, world
        self.timeout = float(os.environ.get('PDES_DP_TIMEOUT_S', '30')) if timeout is None else float(timeout)
        self.on_timeout = on_timeout or self._die
        self.poll = poll
        self._pending = collections.deque()
        self._lock = threading.Lock()
        self._stop = threading.Event()
        self.fired = None
        self._thread = None
        if self.timeout > 0:
            self._thread = threading.Thread(target=self._run, name='pdes-collective-watchdog', daemon=True)
            self._thread.start()

    def arm(self, what, done):
        if self._thread is not None:
            import time
            with self._lock:
                self._pending.append((time.monotonic(), what, done))
                while len(self._pending) > 64:                   # (completed long ago: the oldest entries are checked first)
                    self._pending.popleft()

    @staticmethod
    def _die(message):
        sys.stderr.write(message + '\n')
        sys.stderr.flush()
        os._exit(13)

    def _run(self):
        import time
        while not self._stop.wait(self.poll):
            with self._lock:
                while self._pending and self._safe(self._pending[0][2]):
                    self._pending.popleft()
                head = self._pending[0] if self._pending else None
            if head is not None and time.monotonic() - head[0] > self.timeout:
                self.fired = (f'[pde_surrogate_amd] rank {self.rank} of {self.world} on {socket.gethostname()}: {head[1]} has not '
                              f'completed {time.monotonic() - head[0]:.0f} s after it was enqueued (PDES_DP_TIMEOUT_S = {self.timeout:g}): a '
                              'peer rank is gone or did not enter the collective; aborting this rank')
                self.on_timeout(self.fired)
                return

    @staticmethod
    def _safe(done):
        try:
            return bool(done())
        except Exception:                                         # noqa: BLE001  (a failed query is not a completion)
            return False

    def close(self):
        self._stop.set()


def make_direct_rccl(group, device):
    """a DirectRccl over `group` for the per-step gradient exchange, or None (-> torch.distributed.all_reduce): when the
    backend is not nccl, when PDES_DP_DIRECT=0, or when ANY rank failed to build or probe its communicator -- the ranks
    agree on the outcome over torch.distributed, so that all of them take the same path (DirectRccl's constructor fails
    on every rank or on none before it enters RCCL; the probe's verdict is exchanged below)"""
    if os.environ.get('PDES_DP_DIRECT', '1') == '0' or dist.get_backend(group) != 'nccl':
        return None
    device = torch.device(device)
    comm, err = None, None
    try:
        with torch.cuda.device(device):
            comm = DirectRccl(group, device)
            probe = torch.ones(4, device=device)      # one small all-reduce: every rank must read back the number of ranks
            comm.all_reduce_sum_(probe.data_ptr(), 4, torch.cuda.current_stream(device).cuda_stream)
            if float(probe[0].item()) != float(comm.world):
                raise RuntimeError(f'probe all-reduce returned {float(probe[0].item())}, expected {comm.world}')
    except Exception as e:                           # noqa: BLE001  (any failure means: use the torch path, on every rank)
        err = e
    ok = torch.tensor([0.0 if err is not None else 1.0], device=device)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if float(ok.item()) == 1.0:
        return comm
    if comm is not None and err is None:
        comm.close()
    import warnings
    warnings.warn('direct RCCL communicator unavailable (' + (f'{type(err).__name__}: {err}' if err else 'on another rank')
                  + '); the gradient exchange goes through torch.distributed.all_reduce')
    return None


def share_gpu():
    """PDES_DP_SHARE_GPU=1 (tests on a one-GPU box): every rank uses GPU 0 and the group is gloo -- RCCL refuses two ranks on
    one device; the gradient exchange then goes through torch.distributed.all_reduce"""
    return os.environ.get('PDES_DP_SHARE_GPU', '0') == '1'


def local_device(local_rank):
    """the GPU of this rank: its local rank's, or GPU 0 for every rank under PDES_DP_SHARE_GPU=1"""
    return torch.device('cuda', 0 if share_gpu() else local_rank)


def init_from_env(backend=None):
    """torchrun-style rendezvous: returns (rank, local_rank, world_size); world 1 = no process group"""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() and not share_gpu() else 'gloo'
        kw = {}
        if backend == 'nccl':
            torch.cuda.set_device(local)
            kw['device_id'] = torch.device('cuda', local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local, world


def _parse_cpulist(text):
    cpus = []
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def _cpulist_str(cpus):
    cpus, out, i = sorted(cpus), [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else f'{cpus[i]}-{cpus[j]}')
        i = j + 1
    return ','.join(out)


def gpu_numa_cpus(device, sysfs='/sys/bus/pci/devices'):
    """(numa_node, cpus local to it) of a GPU from sysfs: `<sysfs>/<pci address>/{numa_node,local_cpulist}` with the PCI
    address torch reports for the device; (None, []) when the files are not there (containers without sysfs)"""
    try:                                   # (pinning is an optimisation: a torch without these attributes must not abort a run)
        pr = torch.cuda.get_device_properties(device)
        addr = '%04x:%02x:%02x.0' % (getattr(pr, 'pci_domain_id', 0), pr.pci_bus_id, pr.pci_device_id)
    except (AttributeError, RuntimeError):
        return None, []
    try:
        with open(os.path.join(sysfs, addr, 'numa_node')) as f:
            node = int(f.read().strip())
        with open(os.path.join(sysfs, addr, 'local_cpulist')) as f:
            cpus = _parse_cpulist(f.read())
    except (OSError, ValueError):
        return None, []
    return node, cpus


def cpu_core_groups(cpus, sysfs='/sys/devices/system/cpu'):
    """the hardware threads of `cpus` grouped by physical core (`topology/thread_siblings_list`), cores in the order of
    their first thread; without sysfs every CPU is its own core"""
    groups, seen = [], set()
    for c in cpus:
        if c in seen:
            continue
        sib = [c]
        try:
            with open(os.path.join(sysfs, f'cpu{c}', 'topology', 'thread_siblings_list')) as f:
                sib = [t for t in _parse_cpulist(f.read()) if t in set(cpus)] or [c]
        except (OSError, ValueError):
            pass
        sib = [t for t in sib if t not in seen]
        seen.update(sib)
        groups.append(sib)
    return groups


def plan_affinity(local_rank, nodes, cpulists, allowed, core_groups=None):
    """pure part of `pin_rank_to_gpu_numa`: `nodes[r]` / `cpulists[r]` = NUMA node and local CPUs of local rank r's GPU,
    `allowed` = the CPUs this process may run on.  The ranks whose GPUs hang off the same node share that node's PHYSICAL
    cores in contiguous, disjoint slices (rank order) -- a rank gets every hardware thread of its cores (`core_groups`: the
    node's CPUs grouped by core, `cpu_core_groups`; None: every CPU its own core), so that eight enqueueing host threads (+
    their runtime helper threads) neither migrate across sockets nor sit on each other's cores or SMT siblings.
    Returns the CPU list for `local_rank` ([] = leave as is)."""
    mine = [c for c in cpulists[local_rank] if c in allowed]
    if not mine:
        return []
    peers = [r for r in range(len(nodes)) if nodes[r] == nodes[local_rank] and cpulists[r] == cpulists[local_rank]]
    k, n = peers.index(local_rank), len(peers)
    cores = [[c] for c in mine] if core_groups is None else \
        [g for g in ([t for t in grp if t in set(mine)] for grp in core_groups) if g]
    per = len(cores) // n
    if per == 0:
        return mine
    return sorted(t for grp in cores[k * per:(k + 1) * per] for t in grp)


def local_world_size(world=1):
    """ranks on THIS host: LOCAL_WORLD_SIZE of torchrun, else `world` (a single-node launch)"""
    try:
        return max(1, int(os.environ.get('LOCAL_WORLD_SIZE', world)))
    except ValueError:
        return max(1, world)


def pin_rank_to_gpu_numa(device, local_rank=0, local_world=1, group=None):
    """bind this process (every existing thread; new ones inherit) to its share of the cores of its GPU's NUMA node.
    A step is ~0.5 ms of host enqueue work per 1.7 ms of GPU time on EACH rank: a rank whose thread wanders to the other
    socket pays remote-memory latency on every launch packet.  `local_world`: the ranks on this host (`local_world_size`);
    only ranks that report this host's name count as peers.  PDES_PIN=0 disables.  Returns a dict for the logs:
    {'numa_node', 'cpus', 'n_cpus'} or {'pinned': False, 'why': ...}."""
    if os.environ.get('PDES_PIN', '1') == '0':
        return {'pinned': False, 'why': 'PDES_PIN=0'}
    if not hasattr(os, 'sched_setaffinity'):
        return {'pinned': False, 'why': 'no sched_setaffinity on this platform'}
    if torch.device(device).type != 'cuda':
        return {'pinned': False, 'why': 'no GPU'}
    node, cpus = gpu_numa_cpus(device)               # (never raises: every rank reaches the collective below)
    nodes, lists = [node] * max(local_world, 1), [cpus] * max(local_world, 1)
    # (gated on the GROUP's size, which every rank agrees on -- not on this host's rank count: on a heterogeneous multi-node
    #  launch a host with a single rank must still enter the collective the others enter)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        host = socket.gethostname()
        got = [None] * dist.get_world_size(group)
        dist.all_gather_object(got, (host, local_rank, node, cpus), group=group)
        for h, lr, nd, cl in got:                    # peers = the ranks of THIS host, keyed by their local rank
            if h == host and 0 <= lr < local_world:
                nodes[lr], lists[lr] = nd, cl
    if node is None or node < 0 or not cpus:
        return {'pinned': False, 'why': 'no numa_node / local_cpulist in sysfs for the GPU'}
    if local_rank >= len(nodes):
        return {'pinned': False, 'why': f'local rank {local_rank} outside the {len(nodes)} ranks of this host'}
    allowed = os.sched_getaffinity(0)
    take = plan_affinity(local_rank, nodes, lists, allowed, cpu_core_groups(cpus))
    if not take:
        return {'pinned': False, 'why': 'the node\'s CPUs are outside this process\'s allowed set', 'numa_node': node}
    n_threads = set_affinity_all_threads(take)
    return {'pinned': True, 'numa_node': node, 'cpus': _cpulist_str(take), 'n_cpus': len(take), 'threads_moved': n_threads}


def cpu_quota(cgroup_root='/sys/fs/cgroup', proc_cgroup='/proc/self/cgroup'):
    """CPUs' worth of run time per period this process's control group may use (cgroup v2 `cpu.max`, v1
    `cpu.cfs_quota_us / cpu.cfs_period_us`; the tightest limit on the path from the process's group to the root), or None
    when there is no limit.  A container that shows all 256 hardware threads of its host may still own a quota of 16: every
    thread of the group is frozen for the rest of the 100 ms period once the group's threads have used it up."""
    rel = ''
    try:
        with open(proc_cgroup) as f:
            for line in f:
                hid, ctrl, path = line.rstrip('\n').split(':', 2)
                if ctrl == '' or 'cpu' in ctrl.split(','):
                    rel = path.lstrip('/')
                    if ctrl:
                        break
    except (OSError, ValueError):
        pass
    best = None
    parts = [p for p in rel.split('/') if p]
    for depth in range(len(parts), -1, -1):
        for base in (cgroup_root, os.path.join(cgroup_root, 'cpu'), os.path.join(cgroup_root, 'cpu,cpuacct')):
            d = os.path.join(base, *parts[:depth])
            q = None
            try:
                with open(os.path.join(d, 'cpu.max')) as f:
                    a, b = f.read().split()
                    q = None if a == 'max' else float(a) / float(b)
            except (OSError, ValueError):
                try:
                    with open(os.path.join(d, 'cpu.cfs_quota_us')) as f:
                        a = float(f.read())
                    with open(os.path.join(d, 'cpu.cfs_period_us')) as f:
                        b = float(f.read())
                    q = a / b if a > 0 and b > 0 else None
                except (OSError, ValueError):
                    q = None
            if q is not None and (best is None or q < best):
                best = q
    return best


def host_thread_budget(reserve=2, local_world=1):
    """how many compute threads (BLAS / OpenMP pools) this process can run without starving its own enqueueing thread:
    min(CPUs it may run on, its share of the control group's CPU quota among the `local_world` ranks of this host) -
    `reserve`, at least 1"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    q = cpu_quota()
    if q is not None:
        n = min(n, int(q / max(local_world, 1)))
    return max(1, n - reserve)


def limit_host_threads(reserve=2, local_world=1):
    """cap torch's intra-op pool and every BLAS / OpenMP pool threadpoolctl can see at `host_thread_budget`.  Those pools
    size themselves by the hardware threads they SEE (64 OpenBLAS + 128 OpenMP threads on the 256-thread host of an
    MI355X node) and keep spinning for ~100 ms behind their last job; in a container with a CPU quota below that (16 CPUs
    on the boxes this was measured on) the spinning uses the group's quota up, CFS freezes EVERY thread of the group until
    the 100 ms period ends, and the thread that enqueues the training step stands still for 20-40 ms in the middle of
    hipLaunchKernel -- one step in a few hundred takes 25 ms instead of 1.6 (round 6: tools/diag/stall_hunt.py, the group's
    `nr_throttled` counts exactly the runs with such a step).  PDES_HOST_THREADS=0 leaves the pools alone, =N sets N.
    Returns a dict for the logs; idempotent."""
    env = os.environ.get('PDES_HOST_THREADS')
    if env == '0':
        return {'limited': False, 'why': 'PDES_HOST_THREADS=0'}
    q = cpu_quota()
    n = int(env) if env and env.isdigit() else host_thread_budget(reserve, local_world)
    out = {'limited': True, 'cpu_quota': None if q is None else round(q, 2), 'threads': n, 'torch_threads_before': torch.get_num_threads()}
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)
    try:
        import threadpoolctl
        before = {f"{i['internal_api']}:{i.get('prefix')}": i['num_threads'] for i in threadpoolctl.threadpool_info()}
        ctl = threadpoolctl.ThreadpoolController()
        for lib in ctl.lib_controllers:                           # lower, never raise, a pool
            if lib.num_threads is not None and lib.num_threads > n:
                lib.set_num_threads(n)
        out['pools_before'] = before
    except Exception as e:                                        # noqa: BLE001  (an optimisation must not abort a run)
        out['threadpoolctl'] = f'{type(e).__name__}: {e}'[:120]
    return out


def set_affinity_all_threads(cpus):
    """sched_setaffinity for every thread of this process (new threads inherit); returns the number of threads moved"""
    n_threads = 0
    try:
        for tid in os.listdir('/proc/self/task'):
            try:
                os.sched_setaffinity(int(tid), cpus)
                n_threads += 1
            except OSError:
                pass
    except OSError:
        os.sched_setaffinity(0, cpus)
        n_threads = 1
    return n_threads


def shard_indices(perm, step, batch_size, rank, world_size):
    """this rank's minibatch indices of global step `step` (contiguous split of the global batch)"""
    gb = batch_size * world_size
    lo = step * gb + rank * batch_size
    return perm[lo:lo + batch_size]


def allreduce_sum_(flat, group=None):
    """in-place SUM all-reduce of the flat gradient buffer (no-op without a process group)"""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def broadcast_parameters(flat, group=None, src=0):
    """make every rank start from rank `src`'s parameters (belt and braces: seeds already agree)"""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)
    return flat


def broadcast_buffers(model, group=None, src=0):
    """BatchNorm running statistics / num_batches_tracked of rank `src` to everyone (after loading a checkpoint;
    during training they stay rank-local and rank 0's are what a checkpoint stores)"""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        for b in model.buffers():
            dist.broadcast(b, src=src, group=group)
    return model


def mean_over_ranks(values, group=None):
    """list of python floats -> their mean over the ranks (one small all-reduce; identity without a group).
    Used once per epoch for the logged loss terms: every rank holds the mean over ITS shards."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return list(values)
    dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(group) == 'nccl' else torch.device('cpu')
    t = torch.tensor(list(values), dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return (t / dist.get_world_size(group)).cpu().tolist()


def adam_reference_(param, grad, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8,
                    weight_decay=0.0, grad_scale=1.0):
    """torch restatement of the flat HIP Adam kernel (`pdes_adam_step`): used by the CPU tests of the
    data-parallel path and by the GPU test of the kernel; torch.optim.Adam semantics."""
    g = grad * grad_scale
    if weight_decay != 0.0:
        g = g + weight_decay * param
    b1, b2 = betas
    exp_avg.lerp_(g, 1 - b1)
    exp_avg_sq.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2_sqrt = (1 - b2 ** step) ** 0.5
    denom = (exp_avg_sq.sqrt() / bc2_sqrt).add_(eps)
    param.addcdiv_(exp_avg, denom, value=-lr / bc1)
    return param
