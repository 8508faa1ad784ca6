"""one fresh process: data generation + three fused Adam steps of the default net from seed 1; prints checksums (are they the
same in every process?  tools/diag/run_dp.sh runs it a dozen times under different host thread counts)"""
import contextlib, hashlib, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pde_surrogate_amd.models.codec import DenseED
from pde_surrogate_amd.train import MixedResidualTrainer
from pde_surrogate_amd.utils.data import grf_kle_fields
sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:12]
dev = torch.device('cuda:0')
B = 32
d = grf_kle_fields(6 * B, n_kle=64, cache_dir='/tmp')
data = torch.from_numpy(d).to(dev)
torch.manual_seed(1)
with contextlib.redirect_stdout(io.StringIO()):
    net = DenseED(1, 3, 64, blocks=[6, 8, 6]).to(dev).train()
tr = MixedResidualTrainer(net, B, 64, lr=1e-3, device=dev)
init = sha(tr.flat.cpu().numpy())
for s in range(3):
    tr.step(data[s * B:(s + 1) * B], 1e-3)
torch.cuda.synchronize()
print('data', sha(d), 'init', init, 'params after 3 steps', sha(tr.flat.cpu().numpy()), 'host threads', tr.host_threads, flush=True)
