"""The driver reads ONE JSON line from `python bench.py --gpus N --steps K --warmup W`: the keys it needs, their types and
their mutual consistency, checked on the real command (a short run without the extra legs)."""
import json
import math
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    # the driver's own short form: 5 warm-up steps, steps 6-25 of a fresh process timed
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '20', '--warmup', '5',
                        '--no-extras', '--no-cpu-baseline'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines                     # nothing but the result on stdout (RCCL / library banners go to stderr)
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and d['steps'] == 20 and d['warmup'] == 5
    assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert d['unit'] == 'samples/s' and d['dtype'] == 'f32' and 'synthetic' in d['data']
    assert 'samples/sec' in d['metric'] and 'bs=32' in d['metric']
    assert d['config']['workload'].startswith('configs[1]') and d['config']['global_batch'] == 32
    assert 'model' not in d['config']
    assert d['value'] > 1000 and d['ms_per_step'] > 0
    assert math.isclose(d['value'], 32 / (d['ms_per_step'] * 1e-3), rel_tol=2e-3)       # whole-job samples/s of the timed steps
    rf = d['roofline']
    assert rf['bound'] == 'hbm' and rf['unit'] == 'GB/s' and rf['peak'] == 8000.0
    assert math.isclose(rf['frac'], rf['achieved'] / rf['peak'], rel_tol=1e-3) and 0.3 < rf['frac'] < 1.0
    assert rf['traffic'] is None or rf['traffic'] >= 0.99 * rf['algorithmic_bytes_per_launch']
    assert math.isclose(rf['achieved'], rf['algorithmic_bytes_per_launch'] / rf['us_per_launch'] / 1e3, rel_tol=2e-2)
    assert math.isfinite(d['loss_mean_over_run'])
    # a short window lies inside the GPU's clock ramp: the steady state of the same trainer is reported beside it
    # (the contract window must reproduce the steady state.  Round 5's driver run read 2.19 ms against
    #  1.62: this process's BLAS pool threads had used up the container's CPU quota and the enqueueing thread was frozen
    #  for ~25 ms inside the window -- parallel.limit_host_threads; what remains is the clock ramp of the first ~10 steps)
    ss = d['steady_state']
    assert ss['steps'] == 128 and 0.98 * ss['ms_per_step'] < d['ms_per_step'] < 1.10 * ss['ms_per_step'], (d['ms_per_step'], ss)
    # the per-step series: one HIP-event interval per step of the process, the timed ones sum to the timed wall time
    assert len(d['timed_steps_ms']) == 20 and len(d['warmup_steps_ms']) == 5 and len(d['timed_steps_host_enqueue_ms']) == 20
    assert math.isclose(sum(d['timed_steps_ms']), d['ms_per_step'] * 20, rel_tol=0.03)
    assert max(d['timed_steps_ms'][1:]) < 1.25 * ss['ms_per_step'], d['timed_steps_ms']      # no stalled step
    assert d['host_threads']['limited'] and (d['cgroup_cpu_throttled_during_steps'] or {}).get('nr_throttled', 0) == 0


@pytest.mark.gpu
def test_bench_under_torchrun_with_two_ranks_sharing_the_gpu():
    """the WHOLE N > 1 path of bench.py, end to end, on a one-GPU box: `python -m torch.distributed.run --nproc-per-node 2
    bench.py --gpus 2` with PDES_BENCH_SHARE_GPU=1 (both ranks on GPU 0, rendezvous over gloo -- RCCL refuses two ranks
    on one device; the trainer then exchanges through torch.distributed): NUMA pinning of both ranks (disjoint CPU sets),
    the host-bound probe's all-reduce, per-rank step times, the stand-alone exchange timing, ONE JSON line from rank 0
    with the keys the driver reads -- with --global-batch 64 (strong scaling, 32 per rank: one more branch than the weak
    default; one invocation, ~2 minutes: gloo moves every gradient through the host)."""
    import json
    import socket
    import subprocess
    import sys
    root = ROOT
    for extra in (['--global-batch', '64'],):
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        env = dict(os.environ, PDES_BENCH_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '30', '--warmup', '100'] + extra
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith('{')]
        assert len(lines) == 1, p.stdout[-2000:]
        d = json.loads(lines[0])
        assert d['n_gpus'] == 2 and d['ranks'] == 2 and d['steps'] == 30 and d['warmup'] == 100
        assert d['scaling'] == ('strong' if extra else 'weak') and d['config']['global_batch'] == 64
        assert d['value'] > 0 and abs(d['value'] - 64 * 30 / (d['ms_per_step'] * 30 / 1e3)) < 1e-3 * d['value']
        assert len(d['per_rank_ms_per_step']['all']) == 2 and d['per_rank_ms_per_step']['max'] == pytest.approx(d['ms_per_step'], rel=1e-3)
        assert d['exchange_path'] == 'torch.distributed.all_reduce' and d['config']['collective']['backend'] == 'gloo'
        coll = d['config']['collective']
        # the placement of bucket A's all-reduce is probed during the warm-up (20 steps each, MAX over
        # the ranks) and the fastest kept; both are in the line
        assert set(coll['bucket_placement_probe_ms_per_step']) == {'wgrad', 'wgrad_b', 'main'}
        assert coll['bucket_placement'] == min(coll['bucket_placement_probe_ms_per_step'], key=coll['bucket_placement_probe_ms_per_step'].get)
        assert d['allreduce_us_standalone'] > 0 and coll['buckets'] == (1 if coll['bucket_placement'] == 'main' else 2)
        assert len(d['host_affinity']) == 2
        if all(h.get('pinned') for h in d['host_affinity']):
            assert d['host_affinity'][0]['cpus'] != d['host_affinity'][1]['cpus']          # disjoint shares of the node
        assert d['config']['launch_mode_probe'] is not None and 'cpu_baseline' not in d
        assert 'roofline' in d and d['roofline']['frac'] > 0.3
