#!/usr/bin/env python
"""Per-descriptor self-check of the conditional-Glow chain on the GPU: after one training-mode generate() every
descriptor's output is recomputed from the ENGINE'S OWN input buffers with stock torch ops and compared -- the first
line with a large error names the faulty kernel.  (Debugging aid; the parity tests are tests/test_cglow_gpu.py.)
    python tools/archive/debug_glow.py [G18_cglow_small.npz]"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pde_surrogate_amd.models import glow_msc as G       # noqa: E402
from pde_surrogate_amd.models.codec import _get          # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'G18_cglow_small.npz'
    g = np.load(os.path.join(ROOT, 'tests', 'golden', name))
    dev = torch.device('cuda:0')
    lu = any(k.endswith('conv1x1.l') for k in g.files)
    enc = list(g['enc_blocks']) if 'enc_blocks' in g.files else [1, 1, 1]
    flow = list(g['flow_blocks']) if 'flow_blocks' in g.files else [2, 1, 1]
    net = G.MultiScaleCondGlow(16, 1, 3, enc, flow, LUdecompose=lu)
    net.load_state_dict({k[4:]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith('sd0/')})
    net = net.to(dev).train()
    x = torch.from_numpy(g['x']).to(dev)
    eps = [torch.from_numpy(g[f'eps{i}']).to(dev) for i in range(2)]
    if os.environ.get('PDES_CONV_IMPL'):
        print('PDES_CONV_IMPL =', os.environ['PDES_CONV_IMPL'])
    with torch.no_grad():
        y, logp = net.generate(x, eps)
    eng = net._engine(x)
    X = eng.X
    print('y rel', rel(y.cpu(), torch.from_numpy(g['y'])), 'logp', logp.cpu().numpy(), 'ref', g['logp'])
    specs = net._specs
    lp = torch.zeros(x.shape[0], device=dev, dtype=torch.float64)
    for i, s in enumerate(specs):
        src, dst = X[s.src], X[s.dst]
        out = dst[:, s.dst_coff:s.dst_coff + s.cout]
        exp = None
        if s.kind in ('conv', 'raw'):
            inp = src[:, :s.cin]
            if s.kind == 'conv':
                bn = _get(net, s.norm)
                inp = torch.relu(F.batch_norm(inp, None, None, bn.weight, bn.bias, True, 0.0, 1e-5))
            exp = F.conv2d(inp, _get(net, s.conv).weight, None, s.stride, s.pad)
            if i + 1 < len(specs) and specs[i + 1].kind == G.OP_BIAS_SCALE and specs[i + 1].dst == s.dst:
                nx = specs[i + 1]
                exp = exp + G._get_param(net, nx.x['bias']).view(1, -1, 1, 1)
                if nx.x['scale_p']:
                    exp = exp * torch.exp(G._get_param(net, nx.x['scale_p']).view(1, -1, 1, 1) * 3)
        elif s.kind == G.OP_COPY:
            exp = src[:, :s.cin] if not s.x.get('src2') else torch.cat([src[:, :s.cin], X[s.x['src2']]], 1)
        elif s.kind == G.OP_COUPLING:
            h = X[s.x['h']]
            n2 = s.cin // 2
            n1 = s.cin - n2
            sc = torch.sigmoid(h[:, 1::2] + 2.)
            exp = torch.cat([src[:, :n1], src[:, n1:] / sc - h[:, 0::2]], 1)
            lp += sc.log().flatten(1).sum(1).double()
        elif s.kind == G.OP_MIX:
            c, r, npath, cpath = net._meta['mix'][s.x['index']]
            cv, an = _get(net, cpath), _get(net, npath)
            if lu:
                W = cv.p @ ((cv.l * cv.l_mask + cv.eye) @ (cv.u * cv.u_mask + torch.diag(cv.log_s.exp() * cv.sign_s)))
                ld = cv.log_s.sum()
            else:
                W = cv.weight
                ld = torch.det(W.double()).abs().log().float()
            hw = src.shape[2] * src.shape[3]
            exp = (F.conv2d(src, W.view(c, c, 1, 1)) - an.bias) / an.weight
            lp += float((an.weight.abs().log().sum() - ld) * hw)
            off = sum(m[0] * m[0] for m in net._meta['mix'][:s.x['index']])
            print(f'      W table rel {rel(eng.Wtab[off:off + c * c].view(c, c), W):.2e}  logdet {float(eng.logdet[s.x["index"]]):.6f} '
                  f'vs {float((an.weight.abs().log().sum() - ld) * hw):.6f}')
        elif s.kind == G.OP_UNSQUEEZE:
            B, C, H, W_ = src.shape
            exp = src.reshape(B, C // 4, 2, 2, H, W_).transpose(3, 4).reshape(B, C // 4, 2 * H, 2 * W_)
        elif s.kind == G.OP_GAUSS:
            pr = X[s.x['prior']]
            m, l = pr.chunk(2, 1)
            l = l.clamp(-10., float(np.log(5.)))
            e = X[s.x['eps']]
            exp = m + l.exp() * e
            lp += (-0.5 * (float(np.log(2 * np.pi)) + 2 * l + (exp - m) ** 2 / (2 * l).exp())).flatten(1).sum(1).double()
        elif s.kind == G.OP_BIAS_SCALE:
            continue
        print(f'{i:3d} {str(s.kind):5s} {s.src:>6s} -> {s.dst:<6s} cin {s.cin:3d} cout {s.cout:3d} coff {s.dst_coff:3d} '
              f'{tuple(out.shape[2:])}  rel {rel(out, exp):.2e}')
    print('logp (torch ops over the engine buffers)', lp.cpu().numpy(), ' engine', logp.cpu().numpy())


if __name__ == '__main__':
    main()
