"""CPU tests of the data-parallel host logic over gloo, world_size 2 (no GPU needed):
the sharding partitions the global batch, the flat all-reduce + grad_scale = 1/world equals the mean
of per-shard gradients, and every rank ends a step with identical parameters."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pde_surrogate_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, _, w = parallel.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    torch.manual_seed(1)                          # identical init on every rank
    n, B = 64, 8
    param = torch.randn(1000)
    m, v = torch.zeros(1000), torch.zeros(1000)
    parallel.broadcast_parameters(param)
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(5))
    data = torch.arange(n, dtype=torch.float32)
    seen = []
    for step in range(n // (B * world)):
        idx = parallel.shard_indices(perm, step, B, rank, world)
        seen.append(idx.clone())
        # a "gradient" that depends on the shard: mean over the shard of a per-sample vector
        g_local = torch.stack([torch.sin(param * (1 + data[i])) for i in idx]).mean(0)
        g = g_local.clone()
        parallel.allreduce_sum_(g)
        # reference: mean over ALL ranks' shards of the per-shard gradients
        ref = torch.zeros_like(g)
        for rr in range(world):
            ii = parallel.shard_indices(perm, step, B, rr, world)
            ref += torch.stack([torch.sin(param * (1 + data[i])) for i in ii]).mean(0)
        torch.testing.assert_close(g / world, ref / world, rtol=1e-6, atol=1e-6)
        parallel.adam_reference_(param, g, m, v, step + 1, 1e-3, grad_scale=1.0 / world)
    # every rank holds the same parameters after the epoch
    gathered = [torch.zeros_like(param) for _ in range(world)]
    dist.all_gather(gathered, param)
    for t in gathered[1:]:
        assert torch.equal(t, gathered[0])
    out[rank] = torch.cat(seen)
    dist.barrier()
    dist.destroy_process_group()


def test_dp_world2_gloo():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    a, b = out[0], out[1]
    both = torch.cat([a, b])
    assert both.numel() == 64 and len(set(both.tolist())) == 64      # disjoint cover of the epoch


def test_adam_reference_matches_torch_optim():
    torch.manual_seed(0)
    p0 = torch.randn(257)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=3e-3, weight_decay=0.01)
    p, m, v = p0.clone(), torch.zeros(257), torch.zeros(257)
    for step in range(1, 6):
        g = torch.randn(257)
        p_ref.grad = g.clone()
        opt.step()
        parallel.adam_reference_(p, g, m, v, step, 3e-3, weight_decay=0.01)
    torch.testing.assert_close(p, p_ref.detach(), rtol=1e-6, atol=1e-7)


def test_device_loader_shards_partition_epoch():
    from pde_surrogate_amd.utils.load import DeviceLoader
    x = torch.arange(32, dtype=torch.float32).view(32, 1)
    got = []
    for rank in range(2):
        dl = DeviceLoader(x, batch_size=4, device='cpu', seed=3, rank=rank, world_size=2)
        assert len(dl) == 4
        got.append(torch.cat([b[0].flatten() for b in dl]))
    assert sorted(torch.cat(got).tolist()) == list(range(32))


def test_bench_launch_contract_rendezvous_world2_gloo():
    """bench.py under the driver's launch contract (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment,
    --gpus N == WORLD_SIZE), up to the first CUDA call: two processes rendezvous (gloo here, nccl = RCCL on a GPU box),
    count themselves with one all-reduce, rank 0 prints one JSON line; a mismatching --gpus is refused"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                   WORLD_SIZE='2')
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--rendezvous-only'],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    line = json.loads(outs[0][0].strip().splitlines()[-1])
    assert line == {'rendezvous': 'ok', 'ranks': 2, 'world_size': 2, 'backend': 'gloo'}
    assert 'rendezvous' not in outs[1][0]                         # only rank 0 prints the JSON line (gloo itself logs)
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--rendezvous-only'], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and 'WORLD_SIZE=1' in (p.stderr + p.stdout)


def test_bench_global_batch_strong_scaling_split_world2_gloo():
    """bench.py --global-batch 256 (configs[2] strong-scaled) under the launch contract with two ranks:
    per-rank batch = 256 / 2, the two ranks' slices of every global step are disjoint, contiguous and cover the global
    batch; a global batch the ranks do not divide is refused"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                   WORLD_SIZE='2')
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--rendezvous-only',
                                       '--global-batch', '256'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    line = json.loads(outs[0][0].strip().splitlines()[-1])
    assert line['scaling'] == 'strong' and line['global_batch'] == 256 and line['per_rank_batch'] == 128
    r0, r1 = line['first_offsets']
    assert r0 == [0, 256, 512] and r1 == [128, 384, 640]            # ntrain 8192: rank r takes [i*256 + r*128, +128)
    assert [p['pinned'] for p in line['host_affinity']] == [False, False]      # no GPU here: nothing to pin to
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--rendezvous-only',
                        '--global-batch', '255', '--ntrain', '4096'], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0                                         # one rank divides anything
    import bench
    assert bench.batch_offset(31, 256, 7, 32, 8192) == 31 * 256 + 7 * 32
    assert 0 <= bench.batch_offset(40, 256, 7, 32, 8192) <= 8192 - 32         # wraps inside the dataset


def test_affinity_plan_shares_a_numa_node_between_its_ranks():
    """parallel.plan_affinity: eight GPUs on two sockets -> four disjoint slices per socket, inside the allowed set"""
    from pde_surrogate_amd.parallel import plan_affinity, _parse_cpulist, _cpulist_str
    n0 = _parse_cpulist('0-63,128-191')
    n1 = _parse_cpulist('64-127,192-255')
    nodes = [0, 0, 0, 0, 1, 1, 1, 1]
    lists = [n0] * 4 + [n1] * 4
    allowed = set(range(256))
    plans = [plan_affinity(r, nodes, lists, allowed) for r in range(8)]
    assert all(len(p) == 32 for p in plans)
    assert sorted(c for p in plans[:4] for c in p) == sorted(n0) and sorted(c for p in plans[4:] for c in p) == sorted(n1)
    assert _cpulist_str(plans[0]) == '0-31' and _cpulist_str(plans[5]) == '96-127'
    # a container that allows only 8 CPUs of node 0: the node's ranks share what is allowed, node 1's ranks are left alone
    few = set(range(8))
    assert [len(plan_affinity(r, nodes, lists, few)) for r in range(8)] == [2, 2, 2, 2, 0, 0, 0, 0]
    assert plan_affinity(0, [0], [n0], allowed) == n0
    # slices by PHYSICAL core -- with the SMT siblings (c, c + 128) grouped, a rank gets both hardware threads of
    # each of its 16 cores and no two ranks share a core
    groups0 = [[c, c + 128] for c in range(64)]
    by_core = [plan_affinity(r, nodes, lists, allowed, groups0 if r < 4 else [[c, c + 128] for c in range(64, 128)]) for r in range(8)]
    assert _cpulist_str(by_core[0]) == '0-15,128-143' and _cpulist_str(by_core[3]) == '48-63,176-191'
    assert _cpulist_str(by_core[4]) == '64-79,192-207'
    cores_of = lambda plan: {c % 128 for c in plan}
    assert all(len(p_) == 32 and len(cores_of(p_)) == 16 for p_ in by_core)
    assert all(not (cores_of(by_core[a]) & cores_of(by_core[b])) for a in range(4) for b in range(a + 1, 4))


def test_core_groups_and_local_world_size(tmp_path, monkeypatch):
    """parallel.cpu_core_groups reads `topology/thread_siblings_list` (hardware threads of one core stay together; no sysfs:
    every CPU its own core); parallel.local_world_size = torchrun's LOCAL_WORLD_SIZE, not the global world size"""
    from pde_surrogate_amd.parallel import cpu_core_groups, local_world_size
    for c in range(4):
        d = tmp_path / f'cpu{c}' / 'topology'
        d.mkdir(parents=True)
        (d / 'thread_siblings_list').write_text(f'{c % 2},{c % 2 + 2}\n')
    assert cpu_core_groups([0, 1, 2, 3], sysfs=str(tmp_path)) == [[0, 2], [1, 3]]
    assert cpu_core_groups([0, 1], sysfs=str(tmp_path)) == [[0], [1]]            # siblings outside the list are left out
    assert cpu_core_groups([5, 6], sysfs=str(tmp_path / 'absent')) == [[5], [6]]
    monkeypatch.setenv('LOCAL_WORLD_SIZE', '8')
    assert local_world_size(16) == 8
    monkeypatch.delenv('LOCAL_WORLD_SIZE')
    assert local_world_size(4) == 4


def test_mean_over_ranks_and_buffer_broadcast_world2_gloo():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_mean, args=(world, port, out), nprocs=world, join=True)
    assert out[0] == out[1] == [1.5, 15.0]


def _worker_mean(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    parallel.init_from_env(backend='gloo')
    out[rank] = parallel.mean_over_ranks([1.0 + rank, 10.0 * (1 + rank)])
    bn = torch.nn.BatchNorm2d(3)
    bn.running_mean.fill_(float(rank + 1))
    parallel.broadcast_buffers(bn)
    assert float(bn.running_mean[0]) == 1.0
    dist.barrier()
    dist.destroy_process_group()


def test_device_loader_drops_the_last_partial_batch_like_the_reference():
    """utils/load.py:34-35 of the reference: DataLoader(shuffle=True, drop_last=True) -- 22 samples at batch 4 are 5
    batches per epoch; under two ranks (global batch 8) 2 steps, and the ranks' slices never overlap"""
    from pde_surrogate_amd.utils.load import DeviceLoader
    x = torch.arange(22, dtype=torch.float32).reshape(22, 1)
    dl = DeviceLoader(x, batch_size=4, device='cpu', seed=0)
    batches = [b[0].reshape(-1) for b in dl]
    assert len(dl) == 5 and len(batches) == 5 and len(set(torch.cat(batches).tolist())) == 20
    a = DeviceLoader(x, batch_size=4, device='cpu', seed=1, rank=0, world_size=2)
    b = DeviceLoader(x, batch_size=4, device='cpu', seed=1, rank=1, world_size=2)
    assert len(a) == len(b) == 2
    for (ba,), (bb,) in zip(a, b):
        assert not set(ba.reshape(-1).tolist()) & set(bb.reshape(-1).tolist())
    with pytest.raises(ValueError):
        DeviceLoader(x[:3], batch_size=4, device='cpu')


def test_cpu_quota_reads_cgroup_v2_and_v1_and_caps_the_host_thread_pools(tmp_path, monkeypatch):
    """round 6: a container that shows 256 hardware threads but owns a CFS quota of 16 CPUs froze the enqueueing thread when
    this process's BLAS pool threads used the quota up (profiles/r06_a_contract_window_host_stalls.txt):
    parallel.cpu_quota reads the limit, limit_host_threads keeps the pools inside it"""
    # cgroup v2: /proc/self/cgroup "0::/job", limits on the path from the group to the root, the tightest counts
    root = tmp_path / 'cg2'
    (root / 'job').mkdir(parents=True)
    (root / 'cpu.max').write_text('1600000 100000\n')
    (root / 'job' / 'cpu.max').write_text('max 100000\n')
    proc = tmp_path / 'proc2'
    proc.write_text('0::/job\n')
    assert parallel.cpu_quota(str(root), str(proc)) == 16.0
    (root / 'job' / 'cpu.max').write_text('400000 100000\n')
    assert parallel.cpu_quota(str(root), str(proc)) == 4.0
    # cgroup v1: cpu controller mounted under <root>/cpu, quota -1 = unlimited
    root1 = tmp_path / 'cg1'
    (root1 / 'cpu' / 'box').mkdir(parents=True)
    (root1 / 'cpu' / 'box' / 'cpu.cfs_quota_us').write_text('-1\n')
    (root1 / 'cpu' / 'box' / 'cpu.cfs_period_us').write_text('100000\n')
    proc1 = tmp_path / 'proc1'
    proc1.write_text('4:memory:/x\n1:cpu,cpuacct:/box\n0::/\n')
    assert parallel.cpu_quota(str(root1), str(proc1)) is None
    (root1 / 'cpu' / 'box' / 'cpu.cfs_quota_us').write_text('250000\n')
    assert parallel.cpu_quota(str(root1), str(proc1)) == 2.5
    # nothing readable: no limit
    assert parallel.cpu_quota(str(tmp_path / 'none'), str(tmp_path / 'nofile')) is None
    # the budget: min(allowed CPUs, the rank's share of the quota) - reserve, never below 1
    monkeypatch.setattr(parallel, 'cpu_quota', lambda *a: 16.0)
    monkeypatch.setattr(os, 'sched_getaffinity', lambda pid: set(range(256)))
    assert parallel.host_thread_budget(2, 1) == 14 and parallel.host_thread_budget(2, 8) == 1
    monkeypatch.setattr(parallel, 'cpu_quota', lambda *a: None)
    assert parallel.host_thread_budget(2, 1) == 254
    # limit_host_threads lowers (never raises) torch's pool and reports what it did; PDES_HOST_THREADS overrides
    before = torch.get_num_threads()
    import threadpoolctl
    pools = [(lib, lib.num_threads) for lib in threadpoolctl.ThreadpoolController().lib_controllers]
    try:
        monkeypatch.setenv('PDES_HOST_THREADS', '0')
        assert parallel.limit_host_threads() == {'limited': False, 'why': 'PDES_HOST_THREADS=0'}
        monkeypatch.setenv('PDES_HOST_THREADS', '1')
        info = parallel.limit_host_threads()
        assert info['limited'] and info['threads'] == 1 and torch.get_num_threads() == 1
        monkeypatch.setenv('PDES_HOST_THREADS', '64')
        parallel.limit_host_threads()
        assert torch.get_num_threads() == 1                # a cap is never a raise
    finally:
        torch.set_num_threads(before)
        for lib, n in pools:                               # (the rest of the CPU suite keeps its pools)
            if n:
                lib.set_num_threads(n)


def test_reference_epoch_order_draws_the_reference_dataloaders_permutations():
    """round 6: with the reference's seed and creation order (Parser: manual_seed(1) -> DenseED -> loaders) the permutations
    DeviceLoader(order='reference') draws are the ones the reference's own DataLoaders drew in its run of BASELINE
    configs[0] (tests/golden/G25: train epoch 1, test epoch 1, train epoch 2, test epoch 2 -- recorded from
    /root/reference's script by tools/gen_golden.py round6)"""
    import contextlib
    import io
    import numpy as np
    from pde_surrogate_amd.models.codec import DenseED
    from pde_surrogate_amd.utils.load import DeviceLoader, reference_epoch_order
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'G25_config1_cli_run.npz'))
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        DenseED(1, 3, 64, [6, 8, 6], growth_rate=16, init_features=48)       # consumes the generator like the reference's
    for e in range(2):
        assert np.array_equal(reference_epoch_order(512).numpy(), g['train_perms'][e])
        assert np.array_equal(reference_epoch_order(64).numpy(), g['test_perms'][e])
    # through the loader: batches = consecutive slices of that permutation
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        DenseED(1, 3, 64, [6, 8, 6], growth_rate=16, init_features=48)
    x = torch.arange(512, dtype=torch.float32).reshape(512, 1)
    dl = DeviceLoader(x, batch_size=8, device='cpu', order='reference')
    seen = torch.cat([b[0].reshape(-1) for b in dl]).long().numpy()
    assert np.array_equal(seen, g['train_perms'][0])
    with pytest.raises(ValueError):
        DeviceLoader(x, batch_size=8, device='cpu', order='sorted')


def _worker8(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    r, _, w = parallel.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    B, GB, ntrain, n = 32, 256, 8192, 740091                       # configs[2]: global batch 256 = 8 x 32, the default net's buffer
    perm = torch.randperm(ntrain, generator=torch.Generator().manual_seed(1))
    mine = [parallel.shard_indices(perm, step, B, rank, world) for step in range(3)]
    # the flat gradient buffer in the two buckets of the trainer: bucket A = the tail (convolution weights from TransUp1
    # on), bucket B = the head.  'wgrad' / 'wgrad_b': A first (from a weight-gradient stream), then B; 'main': one exchange
    off = 288_651
    g_local = torch.sin(torch.arange(n, dtype=torch.float32) * (rank + 1) * 1e-3)
    expect = sum(torch.sin(torch.arange(n, dtype=torch.float32) * (q + 1) * 1e-3) for q in range(world))
    results = {}
    for place in ('wgrad', 'wgrad_b', 'main'):
        g = g_local.clone()
        if place == 'main':
            dist.all_reduce(g)
        else:
            dist.all_reduce(g[off:])
            dist.all_reduce(g[:off])
        results[place] = g
    assert torch.equal(results['wgrad'], results['wgrad_b'])
    assert torch.allclose(results['wgrad'], results['main'], rtol=0, atol=1e-5) and torch.allclose(results['main'], expect, atol=1e-5)
    # Adam with grad_scale = 1 / world on every rank: identical parameters everywhere
    torch.manual_seed(1)
    p = torch.randn(n)
    parallel.adam_reference_(p, results['wgrad'], torch.zeros(n), torch.zeros(n), 1, 1e-3, grad_scale=1.0 / world)
    digest = float(p.double().sum()), float(p[::997].double().abs().sum())
    # NUMA placement of eight ranks over two sockets with SMT pairs
    n0, n1 = parallel._parse_cpulist('0-63,128-191'), parallel._parse_cpulist('64-127,192-255')
    nodes, lists = [0, 0, 0, 0, 1, 1, 1, 1], [n0] * 4 + [n1] * 4
    groups = [[c, c + 128] for c in (range(64) if rank < 4 else range(64, 128))]
    plan = parallel.plan_affinity(rank, nodes, lists, set(range(256)), groups)
    gathered = [None] * world
    dist.all_gather_object(gathered, ([m.tolist() for m in mine], digest, plan, parallel.host_thread_budget(2, parallel.local_world_size(world))))
    # a collective that never completes on ONE rank: the watchdog names that rank
    if rank == 5:
        msgs = []
        wd = parallel.CollectiveWatchdog(rank, world, timeout=0.3, on_timeout=msgs.append, poll=0.05)
        wd.arm('the gradient all-reduce of training step 7', lambda: False)
        import time
        t0 = time.monotonic()
        while not msgs and time.monotonic() - t0 < 5:
            time.sleep(0.05)
        wd.close()
        out['watchdog'] = msgs[0] if msgs else None
    if rank == 0:
        out['gathered'] = gathered
    dist.barrier()
    dist.destroy_process_group()


def test_dp_world8_gloo():
    """the 8-rank job of BASELINE configs[2] on the CPU: eight processes over gloo -- the shards of
    every global batch of 256 partition it, the two-bucket exchange equals the single one in every placement and the exact
    sum over the ranks, Adam leaves identical parameters on all ranks, the eight NUMA plans are disjoint halves of the two
    sockets by physical core, and a collective that hangs on one rank is reported by that rank's watchdog"""
    world, port = 8, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker8, args=(world, port, out), nprocs=world, join=True)
    got = out['gathered']
    perm = torch.randperm(8192, generator=torch.Generator().manual_seed(1))
    for step in range(3):
        batch = [i for r in range(8) for i in got[r][0][step]]
        assert batch == perm[step * 256:(step + 1) * 256].tolist()                # contiguous slices, rank order, no overlap
        assert all(len(got[r][0][step]) == 32 for r in range(8))
    assert len({got[r][1] for r in range(8)}) == 1                                 # identical parameters after the step
    plans = [got[r][2] for r in range(8)]
    assert all(len(p_) == 32 for p_ in plans) and len({c for p_ in plans for c in p_}) == 256
    assert all({c % 128 for c in plans[a]}.isdisjoint({c % 128 for c in plans[b]}) for a in range(8) for b in range(a + 1, 8))
    assert len({got[r][3] for r in range(8)}) == 1 and got[0][3] >= 1              # every rank: the same pool budget
    msg = out['watchdog']
    assert msg is not None and 'rank 5 of 8' in msg and 'training step 7' in msg and 'PDES_DP_TIMEOUT_S' in msg


def test_bench_rendezvous_world8_global_batch_256_offsets_of_every_rank():
    """`bench.py --gpus 8 --rendezvous-only --global-batch 256` under the driver's launch contract with EIGHT ranks (gloo
    here): per-rank batch 32, and the offsets of every rank into the shared permutation for the first three global steps"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = _free_port()
    procs = []
    for rank in range(8):
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                   WORLD_SIZE='8', LOCAL_WORLD_SIZE='8', OMP_NUM_THREADS='1')
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--rendezvous-only',
                                       '--global-batch', '256'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-500:] for o in outs]
    line = json.loads(outs[0][0].strip().splitlines()[-1])
    assert line['ranks'] == 8 and line['world_size'] == 8 and line['backend'] == 'gloo'
    assert line['scaling'] == 'strong' and line['global_batch'] == 256 and line['per_rank_batch'] == 32
    assert line['first_offsets'] == [[i * 256 + r * 32 for i in range(3)] for r in range(8)]
    assert len(line['host_affinity']) == 8
    assert all('rendezvous' not in o[0] for o in outs[1:])
