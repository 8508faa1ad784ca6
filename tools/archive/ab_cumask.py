#!/usr/bin/env python
"""Two scheduling experiments on one GPU box (same process, interleaved rounds):
  (1) cost of the fork event on the main chain: [small kernel; small kernel] vs [small kernel; hipEventRecord +
      hipStreamWaitEvent(side); small kernel] back to back;
  (2) the weight-gradient stream confined to a subset of the CUs (hipExtStreamCreateWithCUMask) instead of competing
      for all 256: ms per training step for several masks.
    python tools/archive/ab_cumask.py [rounds]"""
import contextlib, ctypes, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pde_surrogate_amd.models.codec import DenseED
from pde_surrogate_amd.train import MixedResidualTrainer
from pde_surrogate_amd.utils.data import grf_kle_fields

dev = torch.device('cuda:0')
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so'))


def event_gap(n=3000):
    a = torch.zeros(256, device=dev)
    side = torch.cuda.Stream(dev)
    ev = [torch.cuda.Event() for _ in range(64)]
    res = {}
    for tag in ('plain', 'event', 'plain', 'event'):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            a.add_(1.0)
            if tag == 'event':
                e = ev[i % 64]
                e.record()
                side.wait_event(e)
            a.add_(1.0)
        torch.cuda.synchronize()
        res.setdefault(tag, []).append((time.perf_counter() - t0) / n * 1e6)
    print('event gap microbench (us per pair of small kernels):', {k: [round(v, 2) for v in vs] for k, vs in res.items()}, flush=True)


def masked_stream(words):
    arr = (ctypes.c_uint32 * len(words))(*words)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def main(rounds=3, steps=150, warm=20, B=32):
    event_gap()
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        model = DenseED(1, 3, 64, [6, 8, 6])
    tr = MixedResidualTrainer(model, B, 64, lr=1e-3, device=dev)
    data = torch.from_numpy(grf_kle_fields(512, cache_dir='/tmp')).to(dev)
    batches = [data[i * B:(i + 1) * B].contiguous() for i in range(512 // B)]
    tr.step(batches[0], 1e-4)
    default = tr.eng._side_stream()
    full = 0xFFFFFFFF
    cfgs = {'default': None,
            'all256': [full] * 8,
            'low128': [full] * 4 + [0] * 4,
            'low192': [full] * 6 + [0] * 2,
            'even128': [0x55555555] * 8,
            'low16of32x8': [0x0000FFFF] * 8,
            'low24of32x8': [0x00FFFFFF] * 8,
            'low96': [full] * 3 + [0] * 5}
    streams = {k: (default if v is None else masked_stream(v)) for k, v in cfgs.items()}
    res = {k: [] for k in cfgs}
    for r in range(rounds):
        for k in cfgs:
            model._side_streams[dev] = streams[k]
            for i in range(warm):
                tr.step(batches[i % len(batches)], 1e-4)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                tr.step(batches[i % len(batches)], 1e-4)
            torch.cuda.synchronize()
            res[k].append((time.perf_counter() - t0) / steps * 1e3)
    for k in cfgs:
        print(f'{k:14s} ' + ' '.join(f'{t:.4f}' for t in res[k]) + f'  | min {min(res[k]):.4f} ms/step', flush=True)


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
