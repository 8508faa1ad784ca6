"""L-BFGS on ONE flat parameter / gradient buffer -- the optimiser of the single-instance solver
(reference solve_conv_mixed_residual.py:124: torch.optim.LBFGS(lr=0.5, max_iter=20, history_size=50), no line search).

torch.optim.LBFGS walks the parameter list (82 tensors: gather the flat gradient, scatter the update) and runs the
two-loop recursion as ~4 * history tiny kernels per iteration, each dot product followed by a host read: 7 ms of host
work per closure evaluation around a 0.9 ms closure.  This class keeps torch's algorithm and stopping rules (the step
below follows torch/optim/lbfgs.py `step` statement by statement, line-search branch excluded) but

  * works in place on the model's flat buffers (`model._flat`, `model._gscratch`: no gather / scatter);
  * keeps the curvature pairs as the rows of one (2 m + 1, n) matrix W = [S; Y; g] in HBM and touches them with TWO
    bandwidth-bound passes per iteration: W @ [s, y, g] (every inner product the update needs, i.e. the Gram matrix of
    the history, maintained incrementally) and coef @ W (the search direction);
  * runs the two-loop recursion on the (2 m + 1)-dimensional COEFFICIENT vectors in fp64 on the host -- exact algebra,
    q and r are linear combinations of {g, s_i, y_i} -- with one device -> host read per iteration (the loss torch reads
    anyway, together with the few inner products).

In exact arithmetic the iterates equal torch.optim.LBFGS's; in floating point the inner products are summed in a
different order (and the small recursion runs in fp64), tests/test_lbfgs_cpu.py pins the agreement.
"""
import numpy as np
import torch


class FlatLBFGS:
    def __init__(self, flat_param, flat_grad, lr=1.0, max_iter=20, max_eval=None, tolerance_grad=1e-7,
                 tolerance_change=1e-9, history_size=100):
        if flat_param.dim() != 1 or flat_param.shape != flat_grad.shape:
            raise ValueError('flat_param and flat_grad must be 1-D tensors of the same length')
        self.x, self.g = flat_param, flat_grad
        self.lr, self.max_iter = float(lr), int(max_iter)
        self.max_eval = max_eval if max_eval is not None else max_iter * 5 // 4
        self.tol_grad, self.tol_change, self.m = float(tolerance_grad), float(tolerance_change), int(history_size)
        n, m = flat_param.numel(), self.m
        kw = dict(device=flat_param.device, dtype=flat_param.dtype)
        self.ld = (n + 3) // 4 * 4                          # rows 16-byte aligned for the HIP inner-product kernel
        self.W = torch.zeros((2 * m + 1, self.ld), **kw)[:, :n]   # rows [0, m): s_i, [m, 2m): y_i (ring slots), 2m: g
        self.V = torch.zeros((3, self.ld), **kw)[:, :n]     # rows s_new, y_new, g: the three vectors of the one pass
        self.on_gpu = flat_param.is_cuda
        if self.on_gpu:
            if flat_param.dtype != torch.float32:
                raise RuntimeError('FlatLBFGS on the GPU works on fp32 buffers (the HIP inner-product kernel)')
            self.nsplit = 16
            self._partials = torch.zeros((self.nsplit, 2 * m + 1, 3), device=flat_param.device, dtype=torch.float64)
            self._pv = torch.zeros((self.nsplit, 3, 3), device=flat_param.device, dtype=torch.float64)
        self.d = torch.zeros(n, **kw)
        self.prev_g = torch.zeros(n, **kw)
        self.slots = []                                     # ring slots in age order (oldest first)
        self.G = np.zeros((2 * m + 1, 2 * m + 1))           # Gram matrix of the rows of W (fp64, host)
        self.H_diag, self.t = 1.0, None
        self.n_iter_total, self.func_evals, self.prev_loss = 0, 0, None
        self.state = {}

    def _products(self):
        """(2m+1+3, 3) fp64 host array: rows of W and the three rows of V against the three rows of V -- ONE read"""
        m = self.m
        if not self.on_gpu:
            return torch.cat([self.W @ self.V.t(), self.V @ self.V.t()], 0).double().cpu().numpy()
        from . import _lib
        L, st = _lib.lib(), _lib.stream_ptr(self.x.device)
        with _lib.device_guard(self.x.device):
            p = self._partials
            rc = L.pdes_multi_dot(self.W.data_ptr(), self.ld, 2 * m + 1, self.V.data_ptr(), 3, self.ld, p.data_ptr(),
                                  self.nsplit, st)
            _lib.check(rc, 'pdes_multi_dot')
            rc = L.pdes_multi_dot(self.V.data_ptr(), self.ld, 3, self.V.data_ptr(), 3, self.ld,
                                  self._pv.data_ptr(), self.nsplit, st)
            _lib.check(rc, 'pdes_multi_dot')
        return torch.cat([p[:, :2 * m + 1].sum(0), self._pv.sum(0)], 0).cpu().numpy()

    # -- host side: the two-loop recursion on coefficient vectors ------------------------------------------------------
    def _direction_coef(self):
        """coefficients c with d = c @ W; torch's two-loop recursion (lbfgs.py: q = -g; alpha_i; r = H q; beta_i)"""
        m, G = self.m, self.G
        c = np.zeros(2 * m + 1)
        c[2 * m] = -1.0                                     # q = -g
        al = {}
        for k in reversed(self.slots):                      # newest first
            ro = 1.0 / G[k, m + k]
            al[k] = ro * float(G[k] @ c)                    # s_k . q
            c[m + k] -= al[k]                               # q -= al * y_k
        c *= self.H_diag                                    # r = q * H_diag
        for k in self.slots:                                # oldest first
            ro = 1.0 / G[k, m + k]
            be = ro * float(G[m + k] @ c)                   # y_k . r
            c[k] += al[k] - be                              # r += (al - be) * s_k
        return c

    @torch.no_grad()
    def step(self, closure):
        """torch.optim.LBFGS.step semantics: returns the loss of the FIRST closure evaluation"""
        m, W, G = self.m, self.W, self.G
        with torch.enable_grad():
            orig_loss = closure()
        loss = float(orig_loss)
        current_evals = 1
        self.func_evals += 1
        if float(self.g.abs().max()) <= self.tol_grad:
            return orig_loss
        n_iter = 0
        while n_iter < self.max_iter:
            n_iter += 1
            self.n_iter_total += 1
            W[2 * m].copy_(self.g)
            if self.n_iter_total == 1:
                self.d.copy_(self.g).neg_()
                self.slots, self.H_diag = [], 1.0
                stats = torch.stack([self.g.dot(self.g), self.g.abs().sum()]).double().cpu().numpy()
                G[2 * m, 2 * m] = stats[0]
                gtd = -stats[0]
            else:
                # candidate pair (y, s) and every inner product with the history in one pass over W
                V = self.V
                torch.mul(self.d, self.t, out=V[0])                    # s = d * t
                torch.sub(self.g, self.prev_g, out=V[1])               # y = g - prev_g
                V[2].copy_(self.g)
                P = self._products()                                   # (2m+1+3, 3): the one host read
                ys, yy = P[2 * m + 1 + 1, 0], P[2 * m + 1 + 1, 1]
                if ys > 1e-10:
                    if len(self.slots) == m:
                        k = self.slots.pop(0)                          # overwrite the oldest pair
                    else:
                        k = len(self.slots)
                    self.slots.append(k)
                    W[k].copy_(V[0])
                    W[m + k].copy_(V[1])
                    # Gram rows / columns of the new s and y against everything currently stored
                    G[k, :], G[:, k] = P[:2 * m + 1, 0], P[:2 * m + 1, 0]
                    G[m + k, :], G[:, m + k] = P[:2 * m + 1, 1], P[:2 * m + 1, 1]
                    G[k, k], G[m + k, m + k] = P[2 * m + 1, 0], yy
                    G[k, m + k] = G[m + k, k] = ys
                    self.H_diag = ys / yy
                # products with the current gradient (row 2m of W was refreshed above)
                G[2 * m, :], G[:, 2 * m] = P[:2 * m + 1, 2], P[:2 * m + 1, 2]
                G[2 * m, 2 * m] = P[2 * m + 1 + 2, 2]
                if ys > 1e-10:                                         # the new pair's products with g
                    G[2 * m, k] = G[k, 2 * m] = P[2 * m + 1 + 2, 0]
                    G[2 * m, m + k] = G[m + k, 2 * m] = P[2 * m + 1 + 2, 1]
                c = self._direction_coef()
                torch.mv(W.t(), torch.as_tensor(c, dtype=W.dtype, device=W.device), out=self.d)
                gtd = float(G[2 * m] @ c)                              # g . d
            self.prev_g.copy_(self.g)
            self.prev_loss = loss
            if self.n_iter_total == 1:
                self.t = min(1.0, 1.0 / float(stats[1])) * self.lr
            else:
                self.t = self.lr
            if gtd > -self.tol_change:
                break
            self.x.add_(self.d, alpha=self.t)
            ls_func_evals = 0
            opt_cond = False
            if n_iter != self.max_iter:
                with torch.enable_grad():
                    loss_t = closure()
                # one read: the loss, max |g| and max |d t| (torch reads them one by one)
                r = torch.stack([loss_t.detach().reshape(()).to(self.g.dtype), self.g.abs().max(),
                                 self.d.abs().max()]).double().cpu().numpy()
                loss = float(r[0])
                opt_cond = r[1] <= self.tol_grad
                dmax = r[2] * abs(self.t)
                ls_func_evals = 1
            else:
                dmax = None
            current_evals += ls_func_evals
            self.func_evals += ls_func_evals
            if n_iter == self.max_iter:
                break
            if current_evals >= self.max_eval:
                break
            if opt_cond:
                break
            if dmax is not None and dmax <= self.tol_change:
                break
            if abs(loss - self.prev_loss) < self.tol_change:
                break
        return orig_loss
