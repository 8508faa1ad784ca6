"""The L-BFGS closure of the single-instance solver (reference solve_conv_mixed_residual.py:131-149) as one launch
sequence -- eager on three streams (default) or, with `use_graph=True`, ONE serial hipGraph replay.

At B = 1 the closure is ~100 kernels of a few microseconds each (Decoder forward, fused Sobel + nonlinear Darcy
residual, backward): pure launch latency when issued one by one through autograd.  `ResidualClosure` runs the same
arithmetic through the C ABI on the model's primary engine (no autograd graph, gradients land in the flat buffer the
parameters' `.grad` views alias); `torch.optim.LBFGS` drives it unchanged (`optimizer.step(closure)`: it only needs the
returned loss and `p.grad`).  Round 4 (`tools/bench_solver.py`, fresh processes): eager 1,316 closure evaluations per
second / 45.4 FlatLBFGS epochs, one hipGraph 1,218 / 43.6 -- the graph serialises the weight gradients the eager form
runs on the two side streams; it was the faster form (1,208 vs 1,102) while late-built engines drew pool streams that
shared hardware queues.  The graph remains for host-constrained callers (0.02 instead of 0.45 ms of host time).
"""
import torch

from . import _lib


class ResidualClosure:
    def __init__(self, model, latent, perm, weight_bound=10.0, nonlinear=False, beta1=0.0, beta2=0.0, use_graph=False):
        _lib.require_cuda(latent, perm)
        self.model, self.dev = model, latent.device
        self.wb, self.nl, self.b1, self.b2 = float(weight_bound), bool(nonlinear), float(beta1), float(beta2)
        self.eng = model._engine(latent)
        self.eng.reserved = True                          # autograd forwards of the same model get other engines
        self.x = self.eng.X['in']
        self.x.copy_(latent)
        self.K = perm.detach().contiguous().clone()
        B, _, H, W = self.K.shape
        self.B, self.n = B, H
        self.grad_y = torch.empty((B, 3, H, W), device=self.dev)
        self.partials = torch.empty((_lib.loss_partial_rows(B, H, W, 1 if nonlinear else 0), 4), device=self.dev)
        self.terms = torch.zeros(5, device=self.dev)      # {total, const, cont, dirichlet, neumann} of the last call
        self.gflat = model._gscratch
        for p, off in zip(model._params, model._offsets):  # .grad aliases the flat gradient buffer (what LBFGS gathers)
            p.grad = self.gflat[off:off + p.numel()].view(p.shape)
        self.use_graph, self._graph = use_graph, None
        self.n_calls = 0
        self._L = _lib.lib()

    def _compute(self):
        st = _lib.stream_ptr(self.dev)
        y = self.eng.forward(self.x, True)
        rc = self._L.pdes_darcy_loss(self.eng.ctx, self.K.data_ptr(), y.data_ptr(), self.grad_y.data_ptr(),
                                     self.partials.data_ptr(), self.terms.data_ptr(), self.B, self.n, self.n,
                                     1.0, 1.0, self.wb, self.wb, 1 if self.nl else 0, self.b1, self.b2, st)
        _lib.check(rc, 'pdes_darcy_loss')
        self.gflat.zero_()
        self.model._grad_dirty = True
        self.eng.backward(self.grad_y)

    def _capture(self):
        bufs = list(self.model.buffers())
        snap = [b.clone() for b in bufs]
        s = torch.cuda.Stream(device=self.dev)
        s.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(s):                        # warm-up outside the capture (lazy initialisation, planner)
            self._compute()
        torch.cuda.current_stream(self.dev).wait_stream(s)
        torch.cuda.synchronize(self.dev)
        for b, v in zip(bufs, snap):                      # the warm-up is not a closure evaluation: undo its BN bookkeeping
            b.copy_(v)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._compute()
        self._graph = g

    def __call__(self):
        """evaluate loss and gradients at the current parameters; returns the loss (0-dim device tensor)"""
        with _lib.device_guard(self.dev):
            if self.use_graph:
                if self._graph is None:
                    self._capture()
                self._graph.replay()
            else:
                self._compute()
        self.n_calls += 1
        return self.terms[0].clone()      # LBFGS keeps the first evaluation's tensor across later calls
