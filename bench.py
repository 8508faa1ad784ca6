#!/usr/bin/env python
"""bench.py -- mixed-residual training throughput of the HIP path on N MI355X (one process per GPU).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one minibatch of the reference loop body (train_codec_mixed_residual.py:224-240):
DenseED forward, fused Sobel+Darcy loss, backward, [RCCL all-reduce], Adam -- on synthetic
GRF-KLE512 64x64 inputs already resident in HBM, batch 32 per GPU (weak scaling: global batch 32*N).
Prints ONE JSON line on rank 0 with `roofline` (fused Sobel+Darcy-residual kernel, HIP-event timed)
and `cpu_baseline` (the CPU oracle, a port of the reference path, timed on this box's host cores).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOSS_BYTES_FWD_BWD = 7 * 64 * 64 * 4      # read K,u,s1,s2 + write du,ds1,ds2 : 114,688 B / sample
HBM_PEAK_GBPS = 8000.0                      # MI355X HBM3E spec (MI355X_MICROARCH.md)


def loss_kernel_timing(dev, B, iters, warmup=10, burst=False):
    """HIP-event timing of pdes_darcy_loss (fwd+bwd, no finalize) on the stream it is launched on.
    Returns (average us per launch, GB/s[, burst GB/s]).  At B = 16384 the launches are NOT stationary: after an
    idle period the first ~3 launches run at ~6.0 TB/s, the power controller then pulls the chip down (to ~3.5 TB/s
    for a few launches) and it settles at ~5.6 TB/s after ~100 launches (profiles/r01_g_loss_kernel_per_launch.csv), so
    the sustained figure is measured after `warmup` back-to-back launches and the burst figure right after a pause."""
    from pde_surrogate_amd import _lib
    K = torch.exp(0.5 * torch.randn(B, 1, 64, 64, device=dev))
    y = torch.randn(B, 3, 64, 64, device=dev)
    g = torch.empty_like(y)
    part = torch.empty(B, 4, device=dev)
    L, st, ctx = _lib.lib(), _lib.stream_ptr(), _lib.context(dev)

    def run():
        rc = L.pdes_darcy_loss(ctx, K.data_ptr(), y.data_ptr(), g.data_ptr(), part.data_ptr(), None, B, 64, 64,
                               1.0, 1.0, 10.0, 10.0, 0, 0.0, 0.0, st)
        assert rc == 0, rc
    gbs_burst = None
    if burst:
        run()
        torch.cuda.synchronize(dev)
        time.sleep(0.2)
        b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        b0.record()
        for _ in range(3):
            run()
        b1.record()
        torch.cuda.synchronize(dev)
        gbs_burst = LOSS_BYTES_FWD_BWD * B / (b0.elapsed_time(b1) * 1e-3 / 3) / 1e9
    for _ in range(warmup):
        run()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize(dev)
    us = e0.elapsed_time(e1) * 1e3 / iters
    gbs = LOSS_BYTES_FWD_BWD * B / (us * 1e-6) / 1e9
    return (us, gbs, gbs_burst) if burst else (us, gbs)


def cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def physical_cores():
    """physical cores of this host (unique (socket, core) pairs of /proc/cpuinfo; SMT siblings counted once)"""
    cores, phys, cid = set(), None, None
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('physical id'):
                    phys = line.split(':', 1)[1].strip()
                elif line.startswith('core id'):
                    cid = line.split(':', 1)[1].strip()
                elif not line.strip():
                    if phys is not None and cid is not None:
                        cores.add((phys, cid))
                    phys = cid = None
        if phys is not None and cid is not None:
            cores.add((phys, cid))
    except OSError:
        pass
    return len(cores) or (os.cpu_count() or 1)


def cgroup_cpu_stat():
    """nr_periods / nr_throttled / throttled_usec of this process's control group (CFS bandwidth control): a non-zero
    `nr_throttled` over the stepped region means the whole group -- the enqueueing thread included -- stood still"""
    for p in ('/sys/fs/cgroup/cpu.stat', '/sys/fs/cgroup/cpu/cpu.stat', '/sys/fs/cgroup/cpu,cpuacct/cpu.stat'):
        try:
            with open(p) as f:
                d = dict(line.split() for line in f.read().splitlines())
            return {k: int(d[k]) for k in ('nr_periods', 'nr_throttled', 'throttled_usec', 'throttled_time') if k in d}
        except (OSError, ValueError):
            continue
    return {}


AFFINITY_AT_START = os.sched_getaffinity(0) if hasattr(os, 'sched_getaffinity') else None


def unpin_host():
    """the CPU baselines time the box's host cores, not the one NUMA node `pin_host` bound the GPU ranks to: give every thread
    of the process the affinity it started with (the pinned process understated the CPU numbers)"""
    if AFFINITY_AT_START is not None:
        from pde_surrogate_amd import parallel
        parallel.set_affinity_all_threads(AFFINITY_AT_START)


def cpu_baseline(bs, steps=40, warm=2):
    """the CPU oracle (port of the reference's PyTorch-CPU path) on this box's host cores (2 warm-up +
    >= 10 timed full steps at bs = 32 -- 40: ~8 s at the ~160 samples/s of this box, the whole leg 10-15 s of CPU work --
    plus the loss-only forward + backward rate, and bs = 8 for config 1).  The rates
    are the BEST over a sweep of torch.set_num_threads (a 128-thread box runs this small workload fastest on a
    fraction of its cores: VERDICT r2 weak #9); the thread count that wins is reported next to the physical-core count"""
    from oracle import codec as oc, darcy as od, train as ot
    from pde_surrogate_amd.utils.data import grf_kle_fields
    phys, logical = physical_cores(), os.cpu_count() or 1
    # the process may be bound to one NUMA node (pin_host): thread counts beyond the PHYSICAL cores it may run on only
    # oversubscribe them (128 threads on the 64 cores of a node: 1.9 instead of 200 samples/s, 4 minutes of sweep)
    allowed = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else logical
    usable = max(1, min(phys, allowed * phys // max(logical, 1)))
    from pde_surrogate_amd import parallel as _par
    quota = _par.cpu_quota()                              # the container's CFS quota (CPUs per period): more threads only get the group throttled
    if quota is not None:
        usable = max(1, min(usable, int(quota)))
    cands = sorted({t for t in (4, 8, 16, 32, 64, usable) if 1 <= t <= usable})
    x = torch.from_numpy(grf_kle_fields(bs, seed=7, cache_dir='/tmp'))
    y = torch.randn(bs, 3, 64, 64)

    def trainer():
        torch.manual_seed(1)
        return ot.CpuTrainer(oc.densed_init(1, 3, [6, 8, 6], 16, 48), [6, 8, 6])

    def full_rate(tr, xb, n, w):
        for _ in range(w):
            tr.step(xb)
        t0 = time.time()
        for _ in range(n):
            tr.step(xb)
        return xb.shape[0] * n / (time.time() - t0)

    def loss_rate(n):
        for i in range(2 + n):
            if i == 2:
                t1 = time.time()
            yy = y.clone().requires_grad_(True)
            od.mixed_residual_loss(x, yy, 10.0)[0].backward()
        return bs * n / (time.time() - t1)
    tr = trainer()
    sweep = {}
    for t in cands:                                       # short probes: 1 warm-up + 2 timed steps, 20 loss passes
        torch.set_num_threads(t)
        sweep[t] = (round(full_rate(tr, x, 2, 1), 2), round(loss_rate(20), 1))
        if sweep[t][0] < 0.5 * max(v[0] for v in sweep.values()) and sweep[t][1] < 0.5 * max(v[1] for v in sweep.values()):
            break                                         # past the optimum for both rates: more threads only cost time
    cands = sorted(sweep)
    t_full = max(cands, key=lambda t: sweep[t][0])
    t_loss = max(cands, key=lambda t: sweep[t][1])
    torch.set_num_threads(t_full)
    tr = trainer()
    v_full = full_rate(tr, x, steps, warm)
    x8 = x[:8].contiguous()
    v_full8 = full_rate(trainer(), x8, 5, 2)
    torch.set_num_threads(t_loss)
    n_loss = 200
    v_loss = loss_rate(n_loss)
    torch.set_num_threads(t_full)
    return {'value': round(v_full, 2), 'unit': 'samples/s', 'cores': t_full, 'kind': 'port', 'cpu_model': cpu_model(),
            'physical_cores': phys, 'logical_cpus': logical, 'cpus_allowed': allowed, 'physical_cores_usable': usable,
            'cgroup_cpu_quota': None if quota is None else round(quota, 2),
            'loss_only_samples_per_s': round(v_loss, 1), 'loss_only_threads': t_loss,
            'bs8_samples_per_s': round(v_full8, 2),
            'thread_sweep': {str(t): {'full_step_samples_per_s': sweep[t][0], 'loss_only_samples_per_s': sweep[t][1]}
                             for t in cands},
            'sample': f'{steps} full training steps (fwd+loss+bwd+Adam) at bs={bs} after {warm} warm-up steps on {t_full} '
                      f'threads, {n_loss} loss-only fwd+bwd passes at bs={bs} on {t_loss} threads, 5 full steps at bs=8 '
                      f'(config 1); thread counts = the best of a sweep over {cands} (1 + 2 steps, 20 loss passes each); '
                      'PyTorch-CPU fp32 oracle (port of the reference path)'}


def conv1x1_timing(dev, B, iters=200):
    """the 1x1 channel-halving layers (north_star: MFMA utilisation of the 1x1 path): live HIP-event timing of the
    forward kernel of TransDown1.conv1 (144 -> 72 at 32x32) and TransUp1.conv1 (200 -> 100 at 16x16) on their own
    descriptors, utilisation = 2*Cout*Cin*H*W*B / t / 157.3 TF (dense f32 MFMA peak of MI355X)"""
    import contextlib
    import ctypes
    import io
    from pde_surrogate_amd import _lib
    from pde_surrogate_amd.models.codec import DenseED
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        net = DenseED(1, 3, 64, [6, 8, 6]).to(dev).train()
    x = torch.exp(0.5 * torch.randn(B, 1, 64, 64, device=dev))
    with torch.no_grad():
        net(x)
    eng = net._engine(x)
    L, st = _lib.lib(), _lib.stream_ptr()
    out = []
    if not hasattr(eng, '_reduce_n'):
        eng._plan_wgrad_scratch()
    for i, s in enumerate(net._specs):
        if s.k != 1:
            continue
        ref = ctypes.byref(eng.descs[i])
        d = eng.descs[i]
        flop = 2.0 * s.cout * s.cin * d.Hout * d.Wout * B
        row = {'layer': s.conv, 'gemm': f'M={s.cout} K={s.cin} N={d.Hout * d.Wout}x{B}'}
        for tag, fn in (('forward', L.pdes_conv_forward), ('data_gradient', L.pdes_conv_backward_data),
                        ('weight_gradient', L.pdes_conv_backward_weight)):
            for _ in range(20):
                fn(eng.ctx, ref, 1, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn(eng.ctx, ref, 1, st)
            e1.record()
            torch.cuda.synchronize(dev)
            us = e0.elapsed_time(e1) * 1e3 / iters
            row[tag] = {'us_per_launch': round(us, 2), 'achieved': round(flop / us / 1e6, 2), 'frac': round(flop / us / 1e6 / 157.3, 4)}
        row.update(row['forward'])                     # (round-2 keys: the forward kernel's figures at the top level)
        out.append(row)
    return out


def config5_timing(dev, n=300):
    """config 5 of BASELINE.json (single-instance nonlinear solver: Decoder [8,6] + mixed residual at B = 1, the closure of
    solve_conv_mixed_residual.py:131-149 there): closure evaluations per second -- forward + fused nonlinear loss + backward
    + the loss value read back on the host (what L-BFGS needs per evaluation) -- as one hipGraph replay, as the same
    launches issued eagerly, and through autograd on the drop-in modules; plus whole L-BFGS epochs (torch.optim.LBFGS,
    max_iter 20, history 50: its host-side two-loop recursion dominates)"""
    import contextlib
    import io
    from pde_surrogate_amd.models.codec import Decoder
    from pde_surrogate_amd.models.darcy import darcy_mixed_residual_loss
    from pde_surrogate_amd.solver import ResidualClosure
    from pde_surrogate_amd.utils.data import grf_kle_fields
    K = torch.from_numpy(grf_kle_fields(9, n_kle=1024, cache_dir='/tmp')[[8]]).to(dev)       # config 5: GRF KLE1024, idx 8
    torch.manual_seed(0)
    z = (torch.randn(1, 1, 16, 16) * 0.5).to(dev)
    out = {'workload': 'Decoder(1, 3, blocks [8, 6]) 16x16 latent -> 64x64, GRF KLE1024 field idx 8, nonlinear Darcy alpha1 = alpha2 = 0.1, B = 1'}

    def rate(fn, k):
        for _ in range(10):
            float(fn())
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(k):
            float(fn())
        return k / (time.perf_counter() - t0)
    for tag, graph in (('hipgraph', True), ('eager', False)):
        with contextlib.redirect_stdout(io.StringIO()):
            net = Decoder(1, 3, [8, 6]).to(dev).train()
        clo = ResidualClosure(net, z, K, 10.0, True, 0.1, 0.1, use_graph=graph)
        out[f'closure_evals_per_s_{tag}'] = round(rate(clo, n), 1)
        if graph:
            from pde_surrogate_amd.lbfgs import FlatLBFGS
            for name, opt, k in (('torch_optim_lbfgs', torch.optim.LBFGS(net.parameters(), lr=0.5, max_iter=20, history_size=50), 4),
                                 ('flat_lbfgs', FlatLBFGS(net._flat, net._gscratch, lr=0.5, max_iter=20, history_size=50), 12)):
                opt.step(clo)                                 # warm-up epoch
                n0 = clo.n_calls
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(k):
                    opt.step(clo)
                torch.cuda.synchronize(dev)
                dt = time.perf_counter() - t0
                out[f'{name}_epochs_per_s'] = round(k / dt, 2)
                out[f'{name}_closure_evals_per_s'] = round((clo.n_calls - n0) / dt, 1)
    with contextlib.redirect_stdout(io.StringIO()):
        net = Decoder(1, 3, [8, 6]).to(dev).train()

    def autograd_closure():
        net.zero_grad()
        loss = darcy_mixed_residual_loss(K, net(z), 10.0, True, 0.1, 0.1)[0]
        loss.backward()
        return loss
    out['closure_evals_per_s_autograd_dropin'] = round(rate(autograd_closure, n // 3), 1)
    return out


def cglow_timing(dev, steps=60, warm=15, cpu_steps=3):
    """one reverse-KL training step of the default multiscale conditional Glow of
    train_cglow_reverse_kl.py (enc [3, 4, 4], flow [6, 6, 6], 32x32 GRF-KLE100 inputs, bs 32, beta 150, weight_bound 50):
    noise, generate() chain, fused Darcy loss, backward chain, Adam -- samples/s of the fused trainer, with the CPU
    restatement (oracle/glow.py, PyTorch-CPU fp32, same math) timed beside it on a bounded sample"""
    import contextlib
    import io
    import numpy as np
    from pde_surrogate_amd.models.glow_msc import MultiScaleCondGlow
    from pde_surrogate_amd.train import ReverseKLTrainer
    from pde_surrogate_amd.utils.data import grf_kle_fields
    B = 32
    data = torch.from_numpy(grf_kle_fields(4 * B, 32, 100, cache_dir='/tmp')).to(dev)
    torch.manual_seed(1)
    np.random.seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        net = MultiScaleCondGlow(32, 1, 3, [3, 4, 4], [6, 6, 6], LUdecompose=True).to(dev).train()
    tr = ReverseKLTrainer(net, B, 32, lr=1.5e-3, weight_bound=50.0, beta=150.0, device=dev)
    for i in range(warm):
        tr.step(data[(i % 4) * B:(i % 4 + 1) * B], 1e-3)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        tr.step(data[(i % 4) * B:(i % 4 + 1) * B], 1e-3)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    means = tr.epoch_means()
    out = {'workload': 'MultiScaleCondGlow(32, 1, 3, enc [3,4,4], flow [6,6,6], LU) reverse-KL step, bs 32, 32x32, fp32',
           'n_params': net.model_size[0], 'descriptors_per_generate': len(net._specs),
           'samples_per_s': round(B / dt, 1), 'ms_per_step': round(dt * 1e3, 3), 'steps': steps,
           'mean_loss_over_the_run': round(means[0], 3), 'finite': bool(np.isfinite(means[0]))}
    # algorithmic flops of the convolutions (the flow operators -- ActNorm, invertible 1x1, affine coupling, squeeze -- are
    # O(pixels x channels): not counted): forward = sum 2 cin cout k^2 Hout Wout B over the descriptors; a training step
    # runs forward + data gradient + weight gradient of (almost) every layer = 3 x that
    fwd = 0
    for sp in net._specs:
        if sp.kind in ('conv', 'raw'):
            h = 32 * sp.scale[0] // sp.scale[1] // sp.stride
            fwd += 2 * sp.cin * sp.cout * sp.k * sp.k * h * h * B
    out['algorithmic_gflop_per_step'] = {'forward_convolutions': round(fwd / 1e9, 3), 'step_approx_3x': round(3 * fwd / 1e9, 3)}
    out['roofline'] = {'bound': 'mfma', 'peak': 157.3, 'unit': 'TFLOP/s', 'achieved': round(3 * fwd / dt / 1e12, 2),
                       'frac': round(3 * fwd / dt / 157.3e12, 4),
                       'note': 'whole step against the f32 matrix peak: 384 launches on 8x8 .. 32x32 maps, launch-count bound '
                               '(~12 us per dependent kernel), see profiles/r05_*_cglow_*'}
    if cpu_steps:
        from oracle import glow as oglow
        sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
        keys = oglow.param_keys(sd)
        for k in keys:
            sd[k].requires_grad_(True)
        opt = torch.optim.Adam([sd[k] for k in keys], lr=1e-3)
        shapes = oglow.latent_shapes(sd, 3, 32)
        xc = data[:B].cpu()

        def cpu_step():
            t0 = time.perf_counter()
            opt.zero_grad(set_to_none=True)
            eps = [torch.randn((B,) + s) for s in shapes]
            loss = oglow.reverse_kl_loss(sd, xc, eps, 150.0, 50.0, True)[0]
            loss.backward()
            opt.step()
            return time.perf_counter() - t0
        # the same thread sweep as cpu_baseline(): a 128-core box runs this workload fastest on a fraction of its cores
        phys, logical = physical_cores(), os.cpu_count() or 1
        # (8 ... 64 threads: on the 128-core box 16 wins at 0.16 s per step, 64 takes 0.8 s and ALL cores 47 s per step --
        # oversubscription of a workload made of small ops; the sweep stops as soon as a step takes twice the best so far)
        from pde_surrogate_amd import parallel as _par
        quota = _par.cpu_quota()
        cap = logical if quota is None else max(1, min(logical, int(quota)))
        cands = sorted({t for t in (8, 16, 32, 64, cap) if 1 <= t <= cap})
        sweep = {}
        for t in cands:
            torch.set_num_threads(t)
            cpu_step()
            sweep[t] = cpu_step()
            if sweep[t] > 2.0 * min(sweep.values()):
                break
        cands = sorted(sweep)
        best = min(cands, key=lambda t: sweep[t])
        torch.set_num_threads(best)
        times = [cpu_step() for _ in range(cpu_steps)]
        cdt = float(np.mean(times))
        out['cpu_baseline'] = {'value': round(B / cdt, 2), 'unit': 'samples/s', 'cores': best, 'kind': 'port',
                               'physical_cores': phys, 'cpu_model': cpu_model(),
                               'thread_sweep_s_per_step': {str(t): round(v, 3) for t, v in sweep.items()},
                               'sample': f'{cpu_steps} steps at bs {B} on {best} threads (best of a sweep over {cands}, 1 warm-up '
                                         f'+ 1 timed step each), oracle/glow.py on PyTorch-CPU fp32'}
    return out


def rendezvous(gpus):
    """torchrun contract: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment, one process per GPU, backend
    nccl (= RCCL over xGMI) -- gloo only when there is no GPU at all (CPU test of this function).  Returns
    (rank, local_rank, world, device)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != gpus:
        raise SystemExit(f'--gpus {gpus} but WORLD_SIZE={world}: launch with python -m torch.distributed.run '
                         f'--nproc-per-node {gpus} bench.py --gpus {gpus}')
    have_gpu = torch.cuda.is_available()
    # PDES_BENCH_SHARE_GPU=1 (tests only): the ranks share GPU 0 and rendezvous over gloo (RCCL refuses two ranks on one
    # device) -- the whole N > 1 code path of this file on a one-GPU box; the numbers of such a run mean nothing
    share = have_gpu and os.environ.get('PDES_BENCH_SHARE_GPU', '0') == '1'
    dev = torch.device('cuda', 0 if share else local) if have_gpu else torch.device('cpu')
    if have_gpu:
        if not share and local >= torch.cuda.device_count():
            raise SystemExit(f'LOCAL_RANK {local} but only {torch.cuda.device_count()} GPUs are visible')
        torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if have_gpu and not share:
            torch.distributed.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
    return rank, local, world, dev


def batch_offset(i, global_batch, rank, b, ntrain):
    """offset into the shared permutation of rank `rank`'s `b` samples of global step i (contiguous split of the global
    batch, parallel.shard_indices; wraps around the dataset)"""
    return (i * global_batch + rank * b) % (ntrain - b + 1)


def pin_host(dev, local, world):
    """each rank onto its share of the cores of its GPU's NUMA node (parallel.pin_rank_to_gpu_numa); all ranks' plans are
    gathered for the line"""
    from pde_surrogate_amd import parallel
    info = parallel.pin_rank_to_gpu_numa(dev, local, parallel.local_world_size(world))
    if world > 1:
        got = [None] * world
        torch.distributed.all_gather_object(got, info)
        return got
    return [info]


def host_enqueue_timing(trainer, load, k=50):
    """wall time the host needs to ENQUEUE a training step (no synchronisation inside the timed region) against the
    time until the GPU has finished it, and where the host time goes (per-phase perf_counter deltas inside
    MixedResidualTrainer: `forward` = weight packing + pdes_conv_forward, `loss`, `backward` = pdes_backward2 +
    pdes_step_tail, the rest = batch gather, [all-reduce], Adam, Python).  The step is GPU-bound as long as
    host_enqueue_ms_per_step < ms_per_step; the margin is what eight ranks sharing one host can lose."""
    dev = trainer.dev
    for i in range(5):
        load(i)
        trainer.step(None, 1e-4)
    # bursts of `burst` steps behind a synchronise: with hundreds of launches outstanding the runtime makes the host
    # wait for queue slots, and the enqueue time of a long unsynchronised run converges to the GPU's step time
    # (0.61 / 0.71 / 0.97 ms per step for bursts of 20 / 5 / 50 in round 3) -- the figure wanted here is the host's own
    burst, reps = 10, max(k // 10, 1)
    trainer.host_prof = {}
    enq, done = [], []
    for r in range(reps):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(burst):
            load(r * burst + i)
            trainer.step(None, 1e-4)
        t1 = time.perf_counter()
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        enq.append((t1 - t0) / burst)
        done.append((t2 - t0) / burst)
    prof, trainer.host_prof = trainer.host_prof, None
    n = max(prof.get('n', 1), 1)
    ms = lambda v: round(1e3 * v / n, 4)
    enq.sort()
    done.sort()
    t0, t1, t2, k = 0.0, enq[len(enq) // 2], done[len(done) // 2], 1          # medians over the bursts
    return {'host_enqueue_ms_per_step': round(1e3 * (t1 - t0) / k, 4), 'until_gpu_done_ms_per_step': round(1e3 * (t2 - t0) / k, 4),
            'steps': reps * burst, 'method': f'median of {reps} bursts of {burst} steps, each behind a device synchronise',
            'host_ms_by_phase': {'forward_incl_weight_packing': ms(prof.get('forward', 0.0)), 'loss': ms(prof.get('loss', 0.0)),
                                 'backward_incl_step_tail': ms(prof.get('backward', 0.0)),
                                 'trainer_step_total': ms(prof.get('step', 0.0)),
                                 'batch_gather_and_loop': round(1e3 * (t1 - t0) / k - 1e3 * prof.get('step', 0.0) / n, 4)}}


def dp1_rccl_timing(dev, data, perm, B, steps, warmup):
    """the SAME training step through the data-parallel branch on ONE rank: an RCCL (backend nccl) process group of
    world size 1, the bucket hook inside pdes_backward2, both all-reduce launches, work.wait -- what a rank of the
    8-GPU job enqueues per step besides the exchange's wire time.  Returns ms per step (and the host enqueue time)."""
    import contextlib
    import io
    from pde_surrogate_amd.models.codec import DenseED
    from pde_surrogate_amd.train import MixedResidualTrainer
    port = 29600 + os.getpid() % 300
    torch.distributed.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1, device_id=dev)
    try:
        torch.manual_seed(1)
        with contextlib.redirect_stdout(io.StringIO()):
            model = DenseED(1, 3, 64, [6, 8, 6], growth_rate=16, init_features=48)
        tr = MixedResidualTrainer(model, B, 64, lr=1e-3, weight_bound=10.0, device=dev,
                                  process_group=torch.distributed.group.WORLD)
        n = data.shape[0]

        def load(i):
            lo = (i * B) % (n - B + 1)
            tr.load_batch(data, perm[lo:lo + B])
        for i in range(warmup):
            load(i)
            tr.step(None, 1e-3)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(steps):
            load(i)
            tr.step(None, 1e-3)
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        h = host_enqueue_timing(tr, load, 50)
        res_exchange = ('ncclAllReduce by pointer on the weight-gradient / main stream (parallel.DirectRccl)' if tr._rccl is not None
                        else 'torch.distributed.all_reduce')
        tr.close()                                            # the communicator goes before its process group
        return {'dp1_rccl_ms_per_step': round(1e3 * (t2 - t0) / steps, 4),
                'dp1_rccl_host_enqueue_ms_per_step': h['host_enqueue_ms_per_step'], 'steps': steps,
                'exchange': res_exchange,
                'buckets': 2 if tr.overlap_allreduce else 1, 'bucket_a_bytes': int(tr.gflat.numel() - tr._bucket_off) * 4,
                'bytes_per_step': int(tr.gflat.numel()) * 4}
    finally:
        torch.distributed.destroy_process_group()


def segments_timing(dev, data, perm, B, steps=100):
    """the same step replayed as linear hipGraphs joined by one event per network stage (use_graph='segments',
    models/codec.py StepProgram): what the host pays when it must be cheap (0.2 ms instead of 0.5), and what that costs on
    the GPU (coarser release of the weight gradients).  (The per-process A/B is profiles/r03_e_launch_modes_ab.log.)"""
    import contextlib
    import io
    from pde_surrogate_amd.models.codec import DenseED
    from pde_surrogate_amd.train import MixedResidualTrainer
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        model = DenseED(1, 3, 64, [6, 8, 6], growth_rate=16, init_features=48)
    tr = MixedResidualTrainer(model, B, 64, lr=1e-3, weight_bound=10.0, device=dev, use_graph='segments')
    n = data.shape[0]

    def load(i):
        lo = (i * B) % (n - B + 1)
        tr.load_batch(data, perm[lo:lo + B])
    for i in range(20):
        load(i)
        tr.step(None, 1e-3)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        load(i)
        tr.step(None, 1e-3)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    h = host_enqueue_timing(tr, load, 50)
    pr = tr._program
    return {'ms_per_step': round(dt * 1e3, 4), 'host_enqueue_ms_per_step': h['host_enqueue_ms_per_step'],
            'graphs': len(pr.graphs), 'kernel_nodes': int(sum(pr.nodes)), 'operations_per_step': pr.n_ops,
            'segments': [list(s_) for s_ in pr.segments]}


def eval_path_timing(dev, n_val=512, bs=64, passes=8):
    """SURVEY 8(f) rank 1, the reference's `test()` (train_codec_mixed_residual.py:166-206): eval-mode forward of the default
    DenseED at its test batch size (64), the forward-only loss kernel, the NRMSE / R^2 accumulators on the device -- one host
    read per pass over the validation set.  Samples/s over `passes` passes of `n_val` fields (synthetic inputs and targets)."""
    import contextlib
    import io
    import numpy as np
    from pde_surrogate_amd.metrics import TestMetrics
    from pde_surrogate_amd.models.codec import DenseED
    from pde_surrogate_amd.models.darcy import darcy_loss_launch
    from pde_surrogate_amd.utils.data import grf_kle_fields
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        model = DenseED(1, 3, 64, [6, 8, 6], growth_rate=16, init_features=48).to(dev)
    x = torch.from_numpy(grf_kle_fields(n_val, cache_dir='/tmp')).to(dev)
    y = torch.randn(n_val, 3, 64, 64, device=dev)
    model.train()
    with torch.no_grad():                      # running statistics from one training-mode forward, as after an epoch
        model(x[:bs])
    model.eval()
    metrics = TestMetrics(3, dev)
    acc = torch.zeros(5, device=dev, dtype=torch.float64)

    def one_pass():
        metrics.reset()
        acc.zero_()
        with torch.no_grad():
            for lo in range(0, n_val, bs):
                out = model(x[lo:lo + bs])
                terms, _ = darcy_loss_launch(x[lo:lo + bs], out, (1.0, 1.0, 10.0, 10.0), False)
                acc.add_(terms)
                metrics.update(out, y[lo:lo + bs])
        return float(acc[0]), metrics.result(np.ones(3))
    one_pass()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(passes):
        loss, (rel, r2) = one_pass()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    return {'workload': f'test() of the reference: eval forward + forward-only loss + NRMSE / R^2, default DenseED, bs {bs}, '
                        f'{n_val} validation fields, {passes} passes (one host read per pass)',
            'samples_per_s': round(n_val * passes / dt, 1), 'ms_per_batch': round(dt / (passes * (n_val // bs)) * 1e3, 4),
            'finite': bool(np.isfinite(loss) and np.all(np.isfinite(rel)))}


def max_likelihood_timing(dev, B, steps=100, warm=20):
    """SURVEY 8(f) rank 1, train_codec_max_likelihood.py:197-211: the data-driven loop body (the same DenseED, F.mse_loss against
    simulation targets) as the fused step with the MSE launch in place of the Darcy loss (MaxLikelihoodTrainer)"""
    import contextlib
    import io
    from pde_surrogate_amd.models.codec import DenseED
    from pde_surrogate_amd.train import MaxLikelihoodTrainer
    from pde_surrogate_amd.utils.data import grf_kle_fields
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        model = DenseED(1, 3, 64, [6, 8, 6], growth_rate=16, init_features=48)
    tr = MaxLikelihoodTrainer(model, B, 64, lr=1e-3, device=dev)
    x = torch.from_numpy(grf_kle_fields(8 * B, cache_dir='/tmp')).to(dev)
    y = torch.randn(8 * B, 3, 64, 64, device=dev)
    for i in range(warm):
        tr.step(x[(i % 8) * B:(i % 8 + 1) * B], y[(i % 8) * B:(i % 8 + 1) * B], 1e-3)
    tr.epoch_means()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        tr.step(x[(i % 8) * B:(i % 8 + 1) * B], y[(i % 8) * B:(i % 8 + 1) * B], 1e-3)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    m = tr.epoch_means()
    return {'workload': f'train_codec_max_likelihood.py loop body: default DenseED, bs {B}, MSE against (synthetic) targets, fused step',
            'samples_per_s': round(B * steps / dt, 1), 'ms_per_step': round(dt / steps * 1e3, 4), 'steps': steps,
            'mse_mean_over_timed_steps': float(m[0]) if len(m) else None}


def config4_timing(dev, B, ntrain=4096, steps=128, warm=32):
    """configs[3] of BASELINE.json: channelized (two-valued, sharp-interface) 64 x 64 fields, ntrain 4096, bs 32, the default
    DenseED from scratch on the fused step -- the same kernels as the headline (they are data independent), on the input
    family that stresses the Sobel / residual stencils and the BatchNorm statistics (reference pointer:
    train_codec_mixed_residual.py:134-143; parity pinned by G23)"""
    import contextlib
    import io
    from pde_surrogate_amd.models.codec import DenseED
    from pde_surrogate_amd.train import MixedResidualTrainer
    from pde_surrogate_amd.utils.data import channelized_fields
    from pde_surrogate_amd.utils.practices import OneCycleScheduler
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        model = DenseED(1, 3, 64, [6, 8, 6], growth_rate=16, init_features=48)
    tr = MixedResidualTrainer(model, B, 64, lr=1e-3, weight_bound=10.0, device=dev)
    data = torch.from_numpy(channelized_fields(ntrain)).to(dev)
    perm = torch.randperm(ntrain, generator=torch.Generator(device='cpu').manual_seed(1)).to(dev)
    sched = OneCycleScheduler(lr_max=1e-3, div_factor=2.0, pct_start=0.3)
    total = warm + steps

    def step(i):
        lo = (i * B) % (ntrain - B + 1)
        tr.load_batch(data, perm[lo:lo + B])
        tr.step(None, sched.step((i + 1) / total))
    for i in range(warm):
        step(i)
    tr.epoch_means()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(warm, total):
        step(i)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    means = tr.epoch_means()
    return {'workload': 'configs[3]: channelized 64x64 (two-valued fields, sharp interfaces; synthetic, '
                        'utils/data.channelized_fields), ntrain=%d, bs=%d, default DenseED, fused step' % (ntrain, B),
            'samples_per_s': round(B * steps / dt, 1), 'ms_per_step': round(dt / steps * 1e3, 4), 'steps': steps, 'warmup': warm,
            'loss_mean_over_timed_steps': round(means[0], 4)}


def dropin_timing(dev, data, perm, B, steps=100, warm=20, optim_module=None):
    """the reference's loop body VERBATIM (train_codec_mixed_residual.py:224-240) on the drop-in modules: `model(input)` through
    autograd, the three loss functions of models/darcy.py, `loss.backward()`, `torch.optim.Adam`, the per-step
    `loss.item()` -- what a maintainer of the reference gets by changing only the imports, beside the
    fused trainer of the headline"""
    import contextlib
    import io
    from pde_surrogate_amd.models.codec import DenseED
    from pde_surrogate_amd.models.darcy import (conv_constitutive_constraint as constitutive_constraint,
                                                conv_continuity_constraint as continuity_constraint,
                                                conv_boundary_condition as boundary_condition)
    from pde_surrogate_amd.utils.image_gradient import SobelFilter
    from pde_surrogate_amd.utils.practices import OneCycleScheduler, adjust_learning_rate
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        model = DenseED(1, 3, 64, [6, 8, 6], growth_rate=16, init_features=48).to(dev)
    optim = optim_module or torch.optim        # the reference: `import torch.optim as optim` ... `optim.Adam(...)` (:151-152)
    optimizer = optim.Adam(model.parameters(), lr=1e-3, weight_decay=0.0)
    scheduler = OneCycleScheduler(lr_max=1e-3, div_factor=2.0, pct_start=0.3)
    sobel_filter = SobelFilter(64, correct=True, device=dev)
    weight_bound, n, total = 10.0, data.shape[0], warm + steps
    model.train()
    loss_train = 0.

    def body(i):
        nonlocal loss_train
        lo = (i * B) % (n - B + 1)
        input = data[perm[lo:lo + B]]                      # (the DataLoader's batch: a gather of the shuffled indices)
        model.zero_grad()
        output = model(input)
        loss_pde = constitutive_constraint(input, output, sobel_filter) + continuity_constraint(output, sobel_filter)
        loss_dirichlet, loss_neumann = boundary_condition(output)
        loss_boundary = loss_dirichlet + loss_neumann
        loss = loss_pde + loss_boundary * weight_bound
        loss.backward()
        lr = scheduler.step((i + 1) / total)
        adjust_learning_rate(optimizer, lr)
        optimizer.step()
        loss_train += loss.item()
    for i in range(warm):
        body(i)
    torch.cuda.synchronize(dev)
    loss_train = 0.
    cg0 = cgroup_cpu_stat()
    import gc
    # (round 6: a generation-2 collection of the Python GC -- 100-110 ms over this process's heap -- landed inside this leg's
    #  timed loop in two of three runs and moved its mean from 1.97 to 3.0 ms per step.  Collected now, survivors frozen: the
    #  loop's own garbage is young.  Collections of >= 1 ms inside the loop are still reported.)
    gc.collect()
    gc.freeze()
    gc_events = []                                      # collections of the Python GC inside the timed loop

    def on_gc(phase, info, _t=[0.0]):
        if phase == 'start':
            _t[0] = time.perf_counter()
        else:
            gc_events.append({'generation': info['generation'], 'ms': round((time.perf_counter() - _t[0]) * 1e3, 2)})
    gc.callbacks.append(on_gc)
    t0 = time.perf_counter()
    marks = [t0]
    for i in range(warm, total):
        body(i)
        marks.append(time.perf_counter())               # (the body ends in loss.item(): every step is synchronised anyway)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    per_step = sorted(b - a for a, b in zip(marks, marks[1:]))
    gc.callbacks.remove(on_gc)
    cg1 = cgroup_cpu_stat()
    throttle = {k: cg1[k] - cg0.get(k, 0) for k in cg1} if cg1 else None
    # the same loop without the reference's per-step host read of the loss (what that synchronisation costs)
    t1 = time.perf_counter()
    item, k = loss_train, min(steps, 50)
    for i in range(k):
        lo = (i * B) % (n - B + 1)
        input = data[perm[lo:lo + B]]
        model.zero_grad()
        output = model(input)
        loss_pde = constitutive_constraint(input, output, sobel_filter) + continuity_constraint(output, sobel_filter)
        loss_dirichlet, loss_neumann = boundary_condition(output)
        loss = loss_pde + (loss_dirichlet + loss_neumann) * weight_bound
        loss.backward()
        optimizer.step()
    torch.cuda.synchronize(dev)
    dt_nosync = (time.perf_counter() - t1) / k
    return {'loop': 'reference loop body verbatim: model(input) -> constitutive + continuity + boundary loss functions -> '
                    'loss.backward() -> torch.optim.Adam.step() -> loss.item()', 'samples_per_s': round(B * steps / dt, 1),
            'ms_per_step': round(dt / steps * 1e3, 4), 'ms_per_step_without_loss_item': round(dt_nosync * 1e3, 4),
            # (median / slowest single step of the timed loop: one host stall of tens of ms moves the mean of 100 steps by 10-50 %)
            'ms_per_step_median': round(per_step[len(per_step) // 2] * 1e3, 4), 'ms_slowest_step': round(per_step[-1] * 1e3, 2),
            'cgroup_cpu_throttled_during_steps': throttle, 'python_gc_collections_during_steps': [e for e in gc_events if e['ms'] >= 1.0],
            'steps': steps, 'warmup': warm, 'loss_mean_over_timed_steps': round(item / steps, 4),
            'loss_kernel_launches_per_step': 2,
            'optimizer': f'{type(optimizer).__module__}.Adam', 'optimizer_fused': optimizer.param_groups[0].get('fused'),
            # (a plain torch.optim.Adam over a HIP network becomes pde_surrogate_amd.optim.Adam at its first step: global step
            #  pre-hook, PDES_ADAM_AUTO_FUSED; True = the steps ran as one launch of the flat kernel)
            'optimizer_flat_kernel': getattr(optimizer, '_flat_state', None) is not None,
            'note': 'one forward-only + one backward launch of the fused loss kernel per step (the three functions share one '
                    'autograd node; upstream gradients reach the kernel through device memory: no host sync in backward)'}


def allreduce_timing(trainer, iters=50):
    """stand-alone cost of the gradient exchange (both buckets back to back, nothing to overlap with): what the step
    would pay for the all-reduce if it were NOT hidden under the backward pass"""
    trainer.gflat.zero_()
    for _ in range(5):
        trainer.exchange_standalone()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        trainer.exchange_standalone()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=256, help='timed steps (default: two epochs of config 2, SURVEY 8(d))')
    ap.add_argument('--warmup', type=int, default=128, help='untimed steps before them (default: one epoch)')
    ap.add_argument('--batch-size', type=int, default=32, help='per-GPU minibatch')
    ap.add_argument('--global-batch', type=int, default=None,
                    help='STRONG scaling: a fixed global minibatch split over the ranks (configs[2]: 256 = 8 x 32; per-rank '
                         'batch = global / ranks).  Default: weak scaling, --batch-size per rank')
    ap.add_argument('--launch-mode', default='auto', choices=['auto', 'eager', 'forward'],
                    help="auto (default): eager launches unless the warm-up finds the host enqueue time above 0.8 x the "
                         "step time, then the forward pass is replayed as a hipGraph ('forward': less host work per step)")
    ap.add_argument('--ntrain', type=int, default=None,
                    help='dataset size (default: 4096 = configs[1] on one GPU, 8192 = configs[2] under torchrun)')
    ap.add_argument('--graph', action='store_true',
                    help='capture the compute part of the step in a hipGraph (default: eager launches with the weight '
                         'gradients on a second HIP stream, which measured faster than one serial graph)')
    ap.add_argument('--no-graph', action='store_true', help='(default; kept for older command lines)')
    ap.add_argument('--segments', action='store_true',
                    help='replay the step as linear hipGraphs per network stage joined by events (StepProgram)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true',
                    help='skip the roofline_1x1 and config5_solver legs (profiling runs: only the training step and the '
                         'loss-kernel roofline launches)')
    ap.add_argument('--leg', default=None, choices=['cglow'],
                    help='internal: run ONE extra leg in this (fresh) process and print its JSON object')
    ap.add_argument('--rendezvous-only', action='store_true',
                    help='initialise the process group (gloo when there is no GPU), count the ranks with one '
                         'all-reduce, print it and exit: exercises the launch contract without touching a kernel')
    args = ap.parse_args()

    # stdout carries ONE JSON line and nothing else: RCCL prints a version banner to file descriptor 1 when a communicator
    # is created (through C stdio, flushed at exit: it would land after the line).  Everything any library writes to fd 1
    # is sent to stderr; the result line goes to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(real_stdout, (json.dumps(obj) + '\n').encode())

    t_mark = [time.perf_counter()]

    def mark(what):                      # wall time of every leg on stderr (the line itself stays the only thing on stdout)
        now = time.perf_counter()
        sys.stderr.write('[bench] %-28s %7.1f s\n' % (what, now - t_mark[0]))
        sys.stderr.flush()
        t_mark[0] = now

    if args.leg == 'cglow':
        emit(cglow_timing(torch.device('cuda:0'), cpu_steps=0 if args.no_cpu_baseline else 3))
        return
    rank, local, world, dev = rendezvous(args.gpus)
    mark('import + rendezvous')
    if args.ntrain is None:
        args.ntrain = 4096 if world == 1 else 8192         # configs[2]: ntrain 8192, global batch 256 = 8 x 32
    if args.rendezvous_only:
        t = torch.ones(1, device=dev if dev.type == 'cuda' else 'cpu')
        if world > 1:
            torch.distributed.all_reduce(t)
        line = {'rendezvous': 'ok', 'ranks': int(t.item()), 'world_size': world,
                'backend': torch.distributed.get_backend() if world > 1 else None}
        if args.global_batch is not None:                  # the strong-scaled split, as the timed run would take it
            if args.global_batch % world:
                raise SystemExit(f'--global-batch {args.global_batch} is not a multiple of the {world} ranks')
            b = args.global_batch // world
            mine = [batch_offset(i, args.global_batch, rank, b, args.ntrain) for i in range(3)]
            got = [None] * world
            if world > 1:
                torch.distributed.all_gather_object(got, mine)
            else:
                got = [mine]
            line.update({'scaling': 'strong', 'global_batch': args.global_batch, 'per_rank_batch': b, 'first_offsets': got,
                         'host_affinity': pin_host(dev, local, world)})
        if rank == 0:
            emit(line)
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    if dev.type != 'cuda':
        raise SystemExit('bench.py needs an MI355X (no CPU fallback for the HIP path)')

    from pde_surrogate_amd.models.codec import DenseED
    from pde_surrogate_amd.train import MixedResidualTrainer
    from pde_surrogate_amd.utils.data import grf_kle_fields
    from pde_surrogate_amd.utils.practices import OneCycleScheduler
    import contextlib
    import io

    B = args.batch_size
    if args.global_batch is not None:
        if args.global_batch % world:
            raise SystemExit(f'--global-batch {args.global_batch} is not a multiple of the {world} ranks')
        B = args.global_batch // world
    pins = pin_host(dev, local, world)
    # BLAS / OpenMP pools within the control group's CPU quota BEFORE the synthetic fields are generated (numpy's 64
    # OpenBLAS threads spin for ~100 ms behind their last product: on a 16-CPU quota that froze the enqueueing thread for
    # 20-40 ms inside the timed window of one short run in seven, BENCH_r05's 2.19 ms per step; EXPERIMENTS round 6)
    from pde_surrogate_amd import parallel as _par
    host_threads = _par.limit_host_threads(local_world=_par.local_world_size(world))
    torch.manual_seed(1)                                   # identical init on every rank
    with contextlib.redirect_stdout(io.StringIO()):
        model = DenseED(1, 3, 64, [6, 8, 6], growth_rate=16, init_features=48)
    trainer = MixedResidualTrainer(model, B, 64, lr=1e-3, weight_bound=10.0, device=dev,
                                   use_graph='segments' if args.segments else args.graph)
    # device-resident synthetic dataset (every rank holds the replica, uses its slice of the global batch)
    # the KLE basis (a 4096x4096 eigen-decomposition, ~20 s) is computed by rank 0 and cached for the others
    if rank == 0:
        fields = grf_kle_fields(args.ntrain, cache_dir='/tmp')
    if world > 1:
        torch.distributed.barrier()
    if rank != 0:
        fields = grf_kle_fields(args.ntrain, cache_dir='/tmp')
    data = torch.from_numpy(fields).to(dev)
    mark('model + synthetic fields')
    gen = torch.Generator(device='cpu').manual_seed(1)
    sched = OneCycleScheduler(lr_max=1e-3, div_factor=2.0, pct_start=0.3)
    total = args.steps + args.warmup
    perm = torch.randperm(args.ntrain, generator=gen).to(dev)
    GB = B * world

    def batch(i):
        lo = batch_offset(i, GB, rank, B, args.ntrain)
        trainer.load_batch(data, perm[lo:lo + B])          # the minibatch gather lands in the trainer's input buffer
        return None

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    # one HIP event behind EVERY step of the process (warm-up and timed): recorded on the stream the step
    # is enqueued on, read only after the timed region's closing synchronise -- the per-step series in the line shows
    # whether a short timed window is steady state, a clock ramp or a one-off stall
    import gc
    gc.collect()                                           # (the heap built up by data generation and construction is collected and
    gc.freeze()                                            #  frozen BEFORE the first step: no GPU work, no step -- a full collection
    #                                                         is ~100 ms in this process, three times the driver's 20-step window)
    gc_events = []                                         # every collection of the Python GC during the stepped region

    def on_gc(phase, info, _t=[0.0]):
        if phase == 'start':
            _t[0] = time.perf_counter()
        else:
            gc_events.append({'generation': info['generation'], 'ms': round((time.perf_counter() - _t[0]) * 1e3, 3),
                              'behind_step': next((j for j in range(total, -1, -1) if step_host[j]), 0)})
    gc.callbacks.append(on_gc)
    if os.environ.get('PDES_BENCH_TRACE', '0') == '1':     # diagnosis: host milliseconds of every step by phase
        trainer.host_prof = {'series': []}
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(total + 1)]
    step_host = [0.0] * (total + 1)                        # host clock when step i's enqueue returned

    def run_step(i, lr):
        trainer.step(batch(i), lr)
        step_ev[i + 1].record()
        step_host[i + 1] = time.perf_counter()

    # warm-up; inside it (auto mode, eager launches) ONE burst of 10 steps is timed behind a device synchronise: how long the
    # host needs to enqueue a step against how long the GPU needs to finish it.  Eight ranks share one host: where the host
    # side comes within 0.8 of the step, the remaining steps replay the forward pass as a hipGraph (bit-identical kernels,
    # ~0.1 ms less host work per step, +0.6 % GPU time on an unconstrained host).  The ranks decide together (MAX).
    probe, i, place_probe = None, 0, None
    can_probe = args.launch_mode == 'auto' and not args.graph and not args.segments and args.warmup >= 16
    cg0 = cgroup_cpu_stat()
    step_ev[0].record()
    step_host[0] = time.perf_counter()
    while i < args.warmup:
        if can_probe and probe is None and i >= min(args.warmup // 2, 24) and args.warmup - i >= 14:
            torch.cuda.synchronize(dev)
            p0 = time.perf_counter()
            for _ in range(10):
                run_step(i, sched.step((i + 1) / total))
                i += 1
            p1 = time.perf_counter()
            torch.cuda.synchronize(dev)
            p2 = time.perf_counter()
            ratio = (p1 - p0) / max(p2 - p0, 1e-9)
            if world > 1:
                t = torch.tensor([ratio], device=dev, dtype=torch.float64)
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                ratio = float(t.item())
            probe = {'host_enqueue_ms': round((p1 - p0) * 100, 4), 'until_gpu_done_ms': round((p2 - p0) * 100, 4),
                     'host_over_step_max_over_ranks': round(ratio, 3), 'threshold': 0.8, 'switched_to': None}
            if ratio > 0.8:
                trainer.set_launch_mode('forward')
                probe['switched_to'] = 'forward'
            continue
        if (world > 1 and place_probe is None and not args.graph and os.environ.get('PDES_DP_BUCKET_STREAM') is None
                and trainer.overlap_allreduce and (probe is not None or not can_probe) and args.warmup - i >= 62):
            # where bucket A's all-reduce goes (MixedResidualTrainer.set_bucket_placement): 20 steps in each placement, the
            # ranks agree on the slowest rank's time (MAX) and keep the fastest placement.  The 8-GPU node is the first
            # place this exchange runs for real: the choice is measured there instead of assumed here.
            place_probe = {}
            for where in trainer.BUCKET_PLACEMENTS:
                try:
                    trainer.set_bucket_placement(where)
                except ValueError:
                    continue
                for _ in range(2):
                    run_step(i, sched.step((i + 1) / total))
                    i += 1
                sync()
                q0 = time.perf_counter()
                for _ in range(18):
                    run_step(i, sched.step((i + 1) / total))
                    i += 1
                torch.cuda.synchronize(dev)
                t = torch.tensor([time.perf_counter() - q0], device=dev, dtype=torch.float64)
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                place_probe[where] = round(float(t.item()) / 18 * 1e3, 4)
            best = min(place_probe, key=place_probe.get)
            trainer.set_bucket_placement(best)
            continue
        run_step(i, sched.step((i + 1) / total))
        i += 1
    if args.launch_mode == 'forward' and not args.graph and not args.segments:
        trainer.set_launch_mode('forward')
        for i in range(3):                                 # builds the program (the step after next is the first replay)
            trainer.step(batch(i), sched.step(1 / total))
    sync()
    t0 = time.perf_counter()
    for i in range(args.warmup, total):
        run_step(i, sched.step((i + 1) / total))
    sync()
    dt = time.perf_counter() - t0
    step_ms = [step_ev[j].elapsed_time(step_ev[j + 1]) for j in range(total)]
    host_ms = [(step_host[j + 1] - step_host[j]) * 1e3 for j in range(total)]
    gc.callbacks.remove(on_gc)
    cg1 = cgroup_cpu_stat()
    throttle = {k: cg1[k] - cg0.get(k, 0) for k in cg1} if cg1 else None
    host_series, trainer.host_prof = (trainer.host_prof or {}).get('series'), None
    per_rank_ms = [dt / args.steps * 1e3]
    if world > 1:
        t = torch.zeros(world, device=dev, dtype=torch.float64)
        t[rank] = dt
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM)
        per_rank_ms = [float(v) / args.steps * 1e3 for v in t.tolist()]
        dt = float(t.max().item())
    means = trainer.epoch_means()
    mark('warm-up + timed steps')
    # a SHORT timed window (the driver's --steps 20 --warmup 5) starts ~8 ms after the process's first kernels: the GPU reaches
    # its loaded clocks ~10 steps after any pause (tools/step_series.py: 1.78, 1.75, 1.74, 1.73, 1.69 ... 1.64 ms from step
    # 11 on, the same ramp after a 20 ms idle mid-run), so steps 6-25 read ~1 % above the rate an epoch of 256 steps runs at.
    # `value` stays what the contract times; the steady state is reported beside it
    steady = None
    if world == 1 and total < 100:
        n_more = 128
        for j in range(16):
            trainer.step(batch(total + j), sched.step(1.0))
        sync()
        s0 = time.perf_counter()
        for j in range(n_more):
            trainer.step(batch(total + 16 + j), sched.step(1.0))
        sync()
        s1 = time.perf_counter()
        steady = {'ms_per_step': round((s1 - s0) / n_more * 1e3, 4), 'samples_per_s': round(B * n_more / (s1 - s0), 1),
                  'steps': n_more, 'after_steps': total + 16,
                  'note': 'same trainer, the steps right behind the timed ones: the timed window of a short run lies inside the '
                          "GPU's clock ramp (tools/step_series.py, EXPERIMENTS.md round 5)"}
        trainer.epoch_means()
        mark('steady-state window')
    ar_us = allreduce_timing(trainer) if world > 1 else None
    host = None
    if not args.graph:                                     # every rank steps (the all-reduce is collective); rank 0 reports
        host = host_enqueue_timing(trainer, lambda i: batch(i))
    if world > 1:
        torch.distributed.barrier()
    dp1 = None
    if world == 1 and not args.no_extras and not args.graph:
        try:
            dp1 = dp1_rccl_timing(dev, data, perm, B, min(args.steps, 100), 20)
        except Exception as e:                                # the headline line must still be printed
            dp1 = {'dp1_rccl_ms_per_step': None, 'error': f'{type(e).__name__}: {e}'[:300]}

    seg = None
    if world == 1 and not args.no_extras and not args.graph and not args.segments:
        try:
            seg = segments_timing(dev, data, perm, B)
        except Exception as e:
            seg = {'ms_per_step': None, 'error': f'{type(e).__name__}: {e}'[:300]}
    c4, dropin, evalp, mlk = None, None, None, None
    if world == 1 and not args.no_extras and not args.graph:
        try:
            c4 = config4_timing(dev, B, steps=min(args.steps, 128))
        except Exception as e:
            c4 = {'ms_per_step': None, 'error': f'{type(e).__name__}: {e}'[:300]}
        try:
            evalp = eval_path_timing(dev)
        except Exception as e:
            evalp = {'samples_per_s': None, 'error': f'{type(e).__name__}: {e}'[:300]}
        try:
            mlk = max_likelihood_timing(dev, B, steps=min(args.steps, 100))
        except Exception as e:
            mlk = {'samples_per_s': None, 'error': f'{type(e).__name__}: {e}'[:300]}
        try:
            # (a) torch.optim untouched: a plain torch.optim.Adam over the HIP network's parameters is retargeted to the flat
            #     kernel at its first step (global step pre-hook of this build; round 6 first selected fused=True there);
            #     (b) `from pde_surrogate_amd import optim` in place of `import torch.optim as optim`: the same class from
            #     the start
            dropin = dropin_timing(dev, data, perm, B, steps=min(args.steps, 100))
            from pde_surrogate_amd import optim as _poptim
            d2 = dropin_timing(dev, data, perm, B, steps=min(args.steps, 100), optim_module=_poptim)
            dropin['with_optim_import_redirected'] = {k: d2[k] for k in ('samples_per_s', 'ms_per_step', 'ms_per_step_without_loss_item',
                                                                          'optimizer', 'loss_mean_over_timed_steps')}
        except Exception as e:
            dropin = {'ms_per_step': None, 'error': f'{type(e).__name__}: {e}'[:300]}
    mark('host / dp1 / segment / config-4 / drop-in legs')
    if rank == 0:
        # HBM traffic from the PMC counters: the newest profiles/r*_loss_variants_pmc.json (tools/pmc_loss_variants.sh: separate
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per kernel, FETCH_SIZE doubled per the gfx950 note) -- used ONLY when
        # the file was measured on the same sources of these kernels as the library loaded now (source fingerprint) and names
        # the kernel the variant launches; a stale file leaves `traffic` null and says why
        import glob
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import bench_loss
        pmc_variants, pmc_src, pmc_why = {}, None, 'no profiles/r*_loss_variants_pmc.json'
        pmcs = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_loss_variants_pmc.json')))
        if pmcs:
            with open(pmcs[-1]) as f:
                pj = json.load(f)
            if pj.get('source_fingerprint') != bench_loss.source_fingerprint():
                pmc_why = (f'profiles/{os.path.basename(pmcs[-1])} was measured on other sources of the loss kernels (fingerprint '
                           f"{pj.get('source_fingerprint')} != {bench_loss.source_fingerprint()}): refused")
            else:
                pmc_variants = {v: e for v, e in pj['variants'].items()
                                if e.get('kernel') == bench_loss.VARIANTS.get(v, (0, 0, None))[2] and 'hbm_bytes_per_launch' in e}
                pmc_src = (f'profiles/{os.path.basename(pmcs[-1])} (B=16384, separate --pmc passes per kernel, FETCH_SIZE doubled: '
                           f"gfx950; same kernel sources as this library: fingerprint {pj['source_fingerprint']})")
        traffic = pmc_variants.get('loss_fwd_bwd', {}).get('hbm_bytes_per_launch')
        traffic_src = pmc_src if traffic is not None else pmc_why
        us32, gb32 = loss_kernel_timing(dev, B, 200)
        usL, gbL, gbBurst = loss_kernel_timing(dev, 16384, 100, warmup=100, burst=True)
        mark('loss-kernel roofline')
        # SURVEY 8(d)'s other priced kernels: forward-only loss (the test() path), the nonlinear law, the
        # stand-alone Sobel pair and its adjoint -- HBM regime (B = 16,384) beside the cache-resident sizes
        variants = {}
        for v in bench_loss.VARIANTS:
            rows = {}
            for Bv, it, w, label in ((16384, 100, 100, 'hbm regime: working set > 256 MiB Infinity Cache'),
                                     (256, 200, 20, 'cache-resident'), (B, 200, 20, 'cache-resident / launch-bound (training batch)')):
                r = bench_loss.time_variant(v, Bv, it, w, dev)
                rows[str(Bv)] = {'us_per_launch': r['us_per_launch'], 'achieved': r['achieved'], 'frac': r['frac'],
                                 'algorithmic_bytes_per_launch': r['algorithmic_bytes_per_launch'], 'regime': label}
            big = rows['16384']
            e = pmc_variants.get(v, {})
            variants[v] = {'bound': 'hbm', 'kernel': bench_loss.VARIANTS[v][2], 'unit': 'GB/s', 'peak': HBM_PEAK_GBPS,
                           'algorithmic_bytes_per_' + bench_loss.VARIANTS[v][1]: bench_loss.VARIANTS[v][0],
                           'achieved': big['achieved'], 'frac': big['frac'], 'us_per_launch': big['us_per_launch'], 'batch': 16384,
                           'traffic': e.get('hbm_bytes_per_launch'),
                           'traffic_over_algorithmic': None if not e else round(e['traffic_over_algorithmic'], 4),
                           'by_batch': rows}
        mark('loss-path roofline variants')
        out = {
            'metric': 'training samples/sec (64x64 GRF-KLE512, bs=%d per GPU)' % B,
            'value': round(GB * args.steps / dt, 1), 'unit': 'samples/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 4),
            'higher_is_better': True, 'scaling': 'strong' if args.global_batch is not None else 'weak', 'vs_baseline': None,
            'dtype': 'f32',
            'data': 'synthetic GRF-KLE512 (exp. covariance ell=0.25, 512 KLE terms), random-init DenseED',
            'config': {'workload': '%s: GRF KLE512 64x64, ntrain=%d, bs=%d per GPU, DenseED blocks [6,8,6] '
                                   'growth 16 init 48 (740,091 params), fp32, Adam + one-cycle LR'
                                   % ('configs[1]' if world == 1 and args.global_batch is None else
                                      ('configs[2] STRONG-scaled (--global-batch %d = %d ranks x %d)' % (GB, world, B)
                                       if args.global_batch is not None else
                                       'configs[2] (global batch %d = %d x %d, weak-scaled points of its 8-GPU run)' % (GB, world, B)),
                                      args.ntrain, B),
                       'global_batch': GB, 'parallelism': 'dp%d' % world, 'hip_graph': bool(args.graph),
                       'launch_mode': 'segments' if args.segments else ('one serial hipGraph' if args.graph else (
                           'forward pass as one hipGraph, backward eager on three streams' if trainer.launch_mode == 'forward'
                           else 'eager, three streams')),
                       'launch_mode_probe': probe,
                       'wgrad_stream': not args.graph,
                       'ranks': torch.distributed.get_world_size() if world > 1 else 1,
                       'collective': None if world == 1 else {
                           'backend': 'nccl (RCCL over xGMI)' if torch.distributed.get_backend() == 'nccl' else torch.distributed.get_backend(), 'bytes_per_step': int(trainer.gflat.numel()) * 4,
                           'exchange': 'ncclAllReduce by pointer on the weight-gradient / main stream (parallel.DirectRccl)'
                                       if trainer._rccl is not None else 'torch.distributed.all_reduce',
                           'buckets': 2 if trainer.overlap_allreduce and trainer.bucket_stream != 'main' and not args.graph else 1,
                           'overlapped_with_backward': bool(trainer.overlap_allreduce and trainer.bucket_stream != 'main' and not args.graph),
                           'allreduce_us_standalone': round(ar_us, 1),
                           'bucket_placement': trainer.bucket_stream,
                           'bucket_placement_probe_ms_per_step': place_probe,
                           'note': 'bucket A (conv weights of the last layers, ~3/4 of the bytes) is all-reduced from the '
                                   'weight-gradient stream inside pdes_backward; allreduce_us_standalone is the exchange '
                                   'timed alone after the run (what the step would pay without the overlap)'},
                       'arithmetic': 'fp32 end to end; convolutions on v_mfma_f32_16x16x4_f32, except the three wide 3x3 '
                                     'layers (196->98: forward, data and weight gradient; the nearest-x2 layers 98->49 and '
                                     '100->100: sub-pixel forward (98->49), data and weight gradients): both operands split '
                                     'into three bf16 terms, six cross products accumulated in fp32 on '
                                     'v_mfma_f32_16x16x32_bf16 (24-bit significand coverage, error vs fp64 equal to the f32 '
                                     'pipe: layer-level adversarial test of all of these kernels; option PDES_MFMA_B3 = 0 '
                                     'put them back on the f32 pipe)'},
            'loss_mean_over_run': round(means[0], 4),
            # GPU time of every step of the process: the interval between the HIP events recorded behind consecutive steps on
            # the launch stream (step W+1's interval contains the barrier + synchronise in front of the timed region), and the
            # host's enqueue time of the same steps.  Long default runs carry the first and last 40 timed steps.
            'timed_steps_ms': [round(v, 4) for v in (step_ms[args.warmup:] if args.steps <= 80 else
                                                     step_ms[args.warmup:args.warmup + 40] + step_ms[-40:])],
            'warmup_steps_ms': [round(v, 4) for v in (step_ms[:args.warmup] if args.warmup <= 80 else
                                                      step_ms[:40] + step_ms[args.warmup - 40:args.warmup])],
            'timed_steps_host_enqueue_ms': [round(v, 4) for v in (host_ms[args.warmup:] if args.steps <= 80 else
                                                                  host_ms[args.warmup:args.warmup + 40] + host_ms[-40:])],
            'timed_steps_event_sum_ms': round(sum(step_ms[args.warmup:]), 4),
            'python_gc_collections_during_steps': [e for e in gc_events if e['generation'] == 2 or e['ms'] > 1.0],
            'host_phase_series_ms_fwd_loss_bwd': host_series,
            'ranks': torch.distributed.get_world_size() if world > 1 else 1,
            'per_rank_ms_per_step': {'min': round(min(per_rank_ms), 4), 'max': round(max(per_rank_ms), 4),
                                     'all': [round(v, 4) for v in per_rank_ms]},
            'exchange_path': None if world == 1 else ('DirectRccl (ncclAllReduce by pointer)' if trainer._rccl is not None
                                                      else 'torch.distributed.all_reduce'),
            'host_affinity': pins,
            'host_threads': host_threads,
            'cgroup_cpu_throttled_during_steps': throttle,
            'allreduce_us_standalone': None if ar_us is None else round(ar_us, 1),
            'host': host,
            'roofline': {'bound': 'hbm', 'kernel': 'darcy_loss_kernel<64,bwd> (fused Sobel+Darcy residual+boundary, fwd+bwd)',
                         'achieved': round(gbL, 1), 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                         'frac': round(gbL / HBM_PEAK_GBPS, 4), 'traffic': traffic, 'traffic_source': traffic_src,
                         'batch': 16384, 'us_per_launch': round(usL, 2),
                         'algorithmic_bytes_per_launch': LOSS_BYTES_FWD_BWD * 16384,
                         'note': 'HBM regime (1.88 GB working set > 256 MiB Infinity Cache), HIP events on the launch stream; '
                                 'sustained = average of 100 launches after 100 back-to-back warm-up launches',
                         'burst_GBps_first_3_launches_after_idle': round(gbBurst, 1),
                         'training_size': {'batch': B, 'us_per_launch': round(us32, 2), 'achieved': round(gb32, 1),
                                           'frac': round(gb32 / HBM_PEAK_GBPS, 4),
                                           'note': 'cache-resident / launch-bound at the training batch size'}},
        }
        out['roofline_variants'] = dict(variants, traffic_source=pmc_src if pmc_variants else pmc_why,
                                        note='HIP events on the launch stream; 16384: 100 launches behind 100 warm-up launches; the '
                                             'stand-alone Sobel kernels process the 3 output channels of the batch (3 x batch planes '
                                             'per launch); 32 / 256 are cache-resident and launch-bound, labelled as such')
        if host is not None:
            out['host_enqueue_ms_per_step'] = host['host_enqueue_ms_per_step']
        if steady is not None:
            out['steady_state'] = steady
        if dp1 is not None:
            out['dp1_rccl_ms_per_step'] = dp1['dp1_rccl_ms_per_step']
            out['dp1_rccl'] = dp1
        if seg is not None:
            out['segment_graphs'] = seg
        if c4 is not None:
            out['config4_channelized'] = c4
        if evalp is not None:
            out['eval_path'] = evalp
        if mlk is not None:
            out['max_likelihood'] = mlk
        if dropin is not None:
            out['dropin'] = dropin
        if world == 1 and not args.no_extras:
            out['roofline_1x1'] = {'bound': 'mfma', 'peak': 157.3, 'unit': 'TFLOP/s',
                                   'kernel': 'conv1x1_mfma_kernel (forward), conv_mfma_kernel<1, ...> (data gradient: the LDS-tiled '
                                             'implicit GEMM, PDES_MFMA_1X1 = 5 since round 5), conv1x1_wgrad_kernel',
                                   'layers': conv1x1_timing(dev, B),
                                   'note': 'the 1x1 channel-halving layers, HIP-event timed stand-alone at the training '
                                           'batch; 0.33-0.68 GFLOP GEMMs: launch / prologue / statistics epilogue bound'}
        if world == 1 and not args.no_extras:
            # in a process of its own, so that a failure of the leg cannot take the headline line with it.  (Round 3 ran it
            # apart because a trainer built late in a process was 14-18 % slower, 5.33 -> 6.30 ms: its side streams came from
            # torch's pool and shared hardware queues -- the weight-gradient streams are per device since round 4,
            # profiles/r04_a_late_trainer_streams.log.)
            import subprocess
            cmd = [sys.executable, os.path.abspath(__file__), '--leg', 'cglow'] + (['--no-cpu-baseline'] if args.no_cpu_baseline else [])
            try:
                # (the child starts with the affinity this process started with, pins itself for its GPU part and unpins
                #  for its CPU baseline)
                r = subprocess.run(cmd, stdout=subprocess.PIPE, timeout=600, check=True,
                                   preexec_fn=(lambda: os.sched_setaffinity(0, AFFINITY_AT_START)) if AFFINITY_AT_START else None)
                out['cglow_reverse_kl'] = json.loads(r.stdout.decode().strip().splitlines()[-1])
            except Exception as e:                      # noqa: BLE001  (the headline line must not depend on an extra leg)
                out['cglow_reverse_kl'] = {'error': f'{type(e).__name__}: {e}'}
        mark('1x1 + cglow legs')
        if world == 1 and not args.no_extras:
            out['config5_solver'] = config5_timing(dev)
            mark('config-5 solver leg')
        if world == 1 and not args.no_cpu_baseline:
            unpin_host()
            out['cpu_baseline'] = cpu_baseline(B)
            mark('cpu_baseline')
        emit(out)
    if world > 1:
        trainer.close()                                       # the RCCL communicator goes before its process group
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
