import sys, os, contextlib, io
sys.path.insert(0, os.getcwd())
import torch
from pde_surrogate_amd.models.codec import Decoder, DenseED
from pde_surrogate_amd.models.darcy import darcy_mixed_residual_loss
dev = torch.device('cuda:0')
torch.manual_seed(3)
dec = Decoder(1, 3, [8, 6]).to(dev).train()
z = (0.5 * torch.randn(1, 1, 16, 16)).to(dev)
K = torch.exp(0.5 * torch.randn(1, 1, 64, 64)).to(dev)
names = [k for k, _ in dec.named_parameters()]
ref = None
bad = 0
for it in range(40):
    dec.zero_grad()
    y = dec(z)
    loss, *_ = darcy_mixed_residual_loss(K, y, 10.0, True, 0.1, 0.1)
    loss.backward()
    norms = torch.stack([p.grad.double().norm() for p in dec.parameters()]).cpu()
    if ref is None:
        ref = norms
    rel = ((norms - ref).abs() / ref)
    if rel.max() > 1e-3:
        bad += 1
        i = int(rel.argmax())
        print('iter', it, 'mismatch', names[i], float(norms[i]), float(ref[i]), 'n_bad_params', int((rel > 1e-3).sum()))
print('bad iterations', bad, 'of 40; ref[0]', float(ref[0]))
