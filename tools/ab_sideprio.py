#!/usr/bin/env python
"""Same-process A/B: the two weight-gradient streams at torch's 'least' priority (0 = normal on this runtime) vs HIP streams
created with hipStreamCreateWithPriority at the runtime's own least priority (hipDeviceGetStreamPriorityRange)."""
import contextlib
import ctypes
import io
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pde_surrogate_amd.models import codec
from pde_surrogate_amd.models.codec import DenseED
from pde_surrogate_amd.train import MixedResidualTrainer
from pde_surrogate_amd.utils.data import grf_kle_fields


def main(rounds=3, steps=150, warm=20, B=32):
    dev = torch.device('cuda:0')
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        model = DenseED(1, 3, 64, [6, 8, 6], growth_rate=16, init_features=48)
    tr = MixedResidualTrainer(model, B, 64, lr=1e-3, weight_bound=10.0, device=dev, use_graph=False)
    data = torch.from_numpy(grf_kle_fields(512, cache_dir='/tmp')).to(dev)
    batches = [data[i * B:(i + 1) * B].contiguous() for i in range(512 // B)]
    for i in range(5):
        tr.step(batches[i], 1e-6)
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so'))
    lo, hi = ctypes.c_int(0), ctypes.c_int(0)
    rc = hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi))
    print('hipDeviceGetStreamPriorityRange rc', rc, 'least', lo.value, 'greatest', hi.value, '| torch:', torch.cuda.Stream.priority_range())
    made = {}
    for prio in sorted({lo.value, 0, hi.value}):
        pair = []
        for _ in range(2):
            h = ctypes.c_void_p()
            rc = hip.hipStreamCreateWithPriority(ctypes.byref(h), 1, prio)        # hipStreamNonBlocking
            assert rc == 0, rc
            pair.append(torch.cuda.ExternalStream(h.value, device=dev))
        made[prio] = pair
    orig = dict(codec._DEVICE_SIDE_STREAMS)
    res = {}
    for r in range(rounds):
        for name in ['torch'] + [f'hip{p}' for p in made]:
            if name == 'torch':
                codec._DEVICE_SIDE_STREAMS.clear(); codec._DEVICE_SIDE_STREAMS.update(orig)
            else:
                a, b = made[int(name[3:])]
                codec._DEVICE_SIDE_STREAMS[dev] = a
                codec._DEVICE_SIDE_STREAMS[(dev, 'b')] = b
            torch.cuda.synchronize()
            for i in range(warm):
                tr.step(batches[i % len(batches)], 1e-6)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                tr.step(batches[i % len(batches)], 1e-6)
            torch.cuda.synchronize()
            res.setdefault(name, []).append((time.perf_counter() - t0) / steps * 1e3)
    for k, v in res.items():
        print(f'{k}: ' + ' '.join(f'{t:.4f}' for t in v) + f'  | min {min(v):.4f} ms/step', flush=True)


if __name__ == '__main__':
    main()
