"""Fused mixed-residual training step on MI355X -- the fast path behind the reference's loop body
(train_codec_mixed_residual.py:224-240): zero_grad, forward, Darcy loss, backward, one-cycle LR,
Adam -- without autograd, host synchronisation or per-parameter kernels.

  * the dataset is device resident; a minibatch is an index_select into a static input buffer;
  * DenseED forward/backward and the fused Sobel+residual loss run through the C ABI;
  * gradients are ONE flat fp32 buffer: data parallelism is one RCCL all-reduce (SUM) of it over
    xGMI, followed by the flat Adam kernel with grad_scale = 1/world_size (BatchNorm stays
    rank-local, exactly what DistributedDataParallel would do);
  * the loss terms are accumulated on the device and read once per epoch (the reference's
    per-step ``loss.item()`` sync, :240, is not needed for its per-epoch mean);
  * the step is replayed as LINEAR hipGraphs joined by one event per backward segment (`use_graph='segments'`,
    models/codec.py StepProgram + csrc/step_graph.hip); `use_graph=True` captures the compute part as ONE serial graph
    (no overlap of the weight gradients: slower), `use_graph=False` launches every kernel eagerly on three streams.
"""
import ctypes
import math
import os
import time

import torch

from . import _lib, parallel
from .models.darcy import darcy_loss_launch  # noqa: F401  (re-exported for callers)


def adam_state_dict(trainer):
    """the fused trainer's Adam moments as a `torch.optim.Adam.state_dict()` of this model's parameters -- what the
    reference stores under 'optimizer_state_dict' (train_cglow_reverse_kl.py:281-289) and feeds to
    `optimizer.load_state_dict` on resume: per-parameter exp_avg / exp_avg_sq / step sliced out of the flat buffers
    (built through a real torch.optim.Adam, so the layout is the installed torch's own)"""
    m = trainer.model
    opt = torch.optim.Adam(m._params, lr=trainer.lr, betas=trainer.betas, eps=trainer.eps, weight_decay=trainer.wd)
    if trainer.step_count > 0:
        for p, off in zip(m._params, m._offsets):
            n = p.numel()
            opt.state[p] = {'step': torch.tensor(float(trainer.step_count)),
                            'exp_avg': trainer.exp_avg[off:off + n].view(p.shape).detach().cpu().clone(),
                            'exp_avg_sq': trainer.exp_avg_sq[off:off + n].view(p.shape).detach().cpu().clone()}
    return opt.state_dict()


def load_adam_state(trainer, sd):
    """restore the flat moments / step count from an 'optimizer_state_dict': a torch.optim.Adam state_dict (the
    reference's, --mode dropin's and, since round 3, --mode fused's) or the flat {'exp_avg', 'exp_avg_sq', 'step'} that
    round-2 fused checkpoints hold.  Returns True when the state was restored; raises on a state of another model."""
    m = trainer.model
    if 'exp_avg' in sd and 'state' not in sd:                        # round-2 fused format
        if sd['exp_avg'].numel() != trainer.exp_avg.numel():
            raise ValueError('optimizer_state_dict belongs to another model (flat moment buffer of another size)')
        trainer.exp_avg.copy_(sd['exp_avg'])
        trainer.exp_avg_sq.copy_(sd['exp_avg_sq'])
        trainer.step_count = int(sd['step'])
        return True
    state = sd.get('state', {})
    if not state:
        return False                                                 # an optimizer that never stepped
    if len(state) != len(m._params):
        raise ValueError(f'optimizer_state_dict has {len(state)} parameter states, the model {len(m._params)} parameters')
    steps = set()
    for i, (p, off) in enumerate(zip(m._params, m._offsets)):
        st = state[i] if i in state else state[str(i)]
        n = p.numel()
        if st['exp_avg'].numel() != n:
            raise ValueError(f'optimizer_state_dict: state {i} has {st["exp_avg"].numel()} elements, parameter {i} {n}')
        trainer.exp_avg[off:off + n].copy_(st['exp_avg'].reshape(-1))
        trainer.exp_avg_sq[off:off + n].copy_(st['exp_avg_sq'].reshape(-1))
        steps.add(int(st['step']))
    if len(steps) != 1:
        raise ValueError(f'optimizer_state_dict: parameters at different step counts {sorted(steps)} (one flat Adam step '
                         'keeps a single count)')
    trainer.step_count = steps.pop()
    return True


def to_torch_adam_state(sd, model, optimizer=None):
    """an 'optimizer_state_dict' in either format -> something torch.optim.Adam(model.parameters()).load_state_dict
    accepts (the flat round-2 format is sliced per parameter; a torch state_dict is returned as is).  The flat format
    carries no hyper-parameters and torch's load_state_dict INSTALLS the param_groups it is handed, so pass the
    `optimizer` that will load the result: its own param_groups (--weight-decay, betas, eps, lr) go into the result."""
    if 'state' in sd or 'exp_avg' not in sd:
        return sd

    class _T:                                   # the four attributes adam_state_dict reads
        pass
    t = _T()
    t.model, t.exp_avg, t.exp_avg_sq, t.step_count = model, sd['exp_avg'], sd['exp_avg_sq'], int(sd['step'])
    t.lr, t.betas, t.eps, t.wd = 1e-3, (0.9, 0.999), 1e-8, 0.0      # placeholders: replaced by the loader's groups below
    out = adam_state_dict(t)
    if optimizer is not None:
        own = optimizer.state_dict()['param_groups']
        if [len(g['params']) for g in own] != [len(g['params']) for g in out['param_groups']]:
            raise ValueError('to_torch_adam_state: the optimizer does not hold the model\'s parameters as one group')
        out['param_groups'] = own
    return out


class MixedResidualTrainer:
    def __init__(self, model, batch_size, imsize=64, lr=1e-3, weight_decay=0.0, weight_bound=10.0,
                 betas=(0.9, 0.999), eps=1e-8, device=None, process_group=None, use_graph=False,
                 nonlinear=False, beta1=0.0, beta2=0.0):
        self.model = model
        self.B, self.n = batch_size, imsize
        self.dev = torch.device(device if device is not None else 'cuda:0')
        if self.dev.type != 'cuda':
            raise RuntimeError('MixedResidualTrainer runs on an MI355X only (no CPU fallback)')
        # the BLAS / OpenMP pools of this process never get more threads than its control group's CPU quota leaves beside
        # the thread that enqueues the step (parallel.limit_host_threads: spinning pool threads got that thread frozen)
        self.host_threads = parallel.limit_host_threads(local_world=parallel.local_world_size(1))
        self.wb = float(weight_bound)
        self.nl, self.nb1, self.nb2 = bool(nonlinear), float(beta1), float(beta2)
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.pg = process_group
        self.world = 1
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
        cin = model._bufs['in'][0]
        model.to(self.dev)
        probe = torch.zeros((batch_size, cin, imsize, imsize), device=self.dev)
        self.eng = model._engine(probe)                   # flattens parameters, allocates buffers
        self.eng.reserved = True                          # autograd forwards of the same model get other engines
        self.ctx = self.eng.ctx
        self.x_static = self.eng.X['in']                  # minibatches are gathered straight into it
        self.x_static.zero_()
        self.flat, self.gflat = model._flat, model._gscratch
        if self.world > 1 or process_group is not None:
            parallel.broadcast_parameters(self.flat, process_group)      # seeds already agree; belt and braces
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.hyper = torch.zeros(8, device=self.dev)              # hipGraph mode: read by the captured Adam kernel
        self._hyper_host = torch.zeros(8, pin_memory=True)
        self._hyper_args = (ctypes.c_float * 8)()                 # eager mode: passed to the kernel by value
        self.step_count = 0
        self.grad_y = torch.empty((batch_size, 3, imsize, imsize), device=self.dev)
        self.partials = torch.empty((_lib.loss_partial_rows(batch_size, imsize, imsize, 1 if nonlinear else 0), 4), device=self.dev)
        self.terms = torch.zeros(5, device=self.dev)
        self.terms_accum = torch.zeros(5, device=self.dev, dtype=torch.float64)
        self.n_accum = 0
        if isinstance(use_graph, int) and not isinstance(use_graph, bool):
            if use_graph not in (0, 1):
                raise ValueError(f'use_graph: {use_graph!r} is neither a bool nor one of the named modes')
            use_graph = bool(use_graph)              # 1 == True passes a membership test but not `is True`
        if not (isinstance(use_graph, bool) or use_graph in ('segments', 'forward')):
            raise ValueError("use_graph: False (eager launches), True (one serial hipGraph), 'segments' (linear graphs per "
                             "stage) or 'forward' (the forward pass + loss as one graph, the backward pass eager)")
        self.segments = use_graph in ('segments', 'forward')
        self.forward_graph = use_graph == 'forward'
        self.use_graph = use_graph is True                    # the single serial graph
        self._program = None
        self.seg_max = int(os.environ.get('PDES_SEG_MAX', '0')) or None
        self._graph = None
        self._hyper_event = torch.cuda.Event() if self.use_graph else None
        self._grad_clean = False          # eager: True once the Adam kernel has cleared the gradient buffer
        self._L = _lib.lib()
        # data parallel: the gradient buffer is exchanged in two buckets.  Bucket A = the convolution weights of the
        # last layers (the tail of the buffer, ~3/4 of the bytes), all-reduced from the weight-gradient stream as soon
        # as pdes_backward has reduced their split-K partials -- it overlaps the rest of the backward pass; bucket B =
        # everything before it, after the end-of-step launch.
        self._bucket_work, self._bucket_off = None, 0
        self._hook = None
        self.host_prof = None             # a dict: host seconds per phase of _step are accumulated into it (bench.py)
        self.overlap_allreduce = os.environ.get('PDES_DP_OVERLAP', '1') != '0'
        # where bucket A's all-reduce is enqueued (set_bucket_placement; PDES_DP_BUCKET_STREAM in the environment):
        #   'wgrad'   on the weight-gradient stream that just reduced its split-K partials (default; the weight gradients
        #             released to that stream afterwards queue behind the communication kernel),
        #   'wgrad_b' on the OTHER weight-gradient stream behind one event (the first stream keeps computing),
        #   'main'    not at the hook at all: the whole buffer in one all-reduce on the main stream after the backward pass
        #             (= PDES_DP_OVERLAP=0: no overlap, nothing shares a stream with the exchange).
        # In every placement the main stream waits for bucket A before bucket B is enqueued: the two collectives of the one
        # communicator are ordered by the streams themselves, not by an assumption about RCCL.
        self.bucket_stream = 'wgrad'
        self._bucket_event = None
        self.set_bucket_placement(os.environ.get('PDES_DP_BUCKET_STREAM', 'wgrad'))
        self._rccl = None
        if self.world > 1 or process_group is not None:
            self._hook_fn = _lib.BUCKET_FN(self._on_bucket)          # keep the callback object alive
            self._hook = _lib.BucketHook(self._hook_fn, None)
            # the per-step exchange: ncclAllReduce enqueued by pointer (parallel.DirectRccl) where the data becomes final:
            # bucket A on the weight-gradient stream right behind the early split-K reduce, the rest on the main stream
            # behind the end-of-step launch -- no extra stream, no events of our own (RCCL orders two collectives of one
            # communicator that are issued on different streams itself; pdes_backward2's final join puts the main stream
            # behind bucket A).  A dedicated high-priority communication stream joined by three events was measured at
            # 5.05 ms per step on one rank (1.75 without the exchange): rejected.  torch.distributed's all_reduce (the only
            # choice over gloo, and the fallback when the direct communicator cannot be made; PDES_DP_DIRECT=0 selects
            # it) costs the host ~0.3 ms per call.
            self._rccl = parallel.make_direct_rccl(process_group, self.dev)
            # a gradient exchange that does not complete within PDES_DP_TIMEOUT_S (30 s) becomes a clear error naming the rank
            self._watchdog = parallel.CollectiveWatchdog(torch.distributed.get_rank(process_group), self.world)

    def close(self):
        """release what the trainer holds outside torch's allocator: the direct RCCL communicator (idempotent)"""
        r, self._rccl = getattr(self, '_rccl', None), None
        w, self._watchdog = getattr(self, '_watchdog', None), None
        if w is not None:
            w.close()
        if r is not None:
            r.close()

    def __del__(self):
        try:
            self.close()
        except Exception:                                    # noqa: BLE001
            pass

    BUCKET_PLACEMENTS = ('wgrad', 'wgrad_b', 'main')

    def set_bucket_placement(self, where):
        """choose BETWEEN steps the stream bucket A's all-reduce is enqueued on (see __init__); bench.py times the three
        during its warm-up under torchrun and keeps the fastest"""
        if where not in self.BUCKET_PLACEMENTS:
            raise ValueError(f'bucket placement {where!r}: one of {self.BUCKET_PLACEMENTS}')
        if where == 'wgrad_b' and getattr(self.model, 'wgrad_streams', 2) != 2:
            raise ValueError("bucket placement 'wgrad_b' needs the second weight-gradient stream (PDES_WGRAD_STREAMS=2)")
        self.bucket_stream = where

    def set_launch_mode(self, use_graph):
        """switch BETWEEN steps among eager launches (False), 'forward' (the forward pass + loss as one hipGraph, eager
        backward) and 'segments' (linear graphs per stage): the three replay the same kernels and are bit-identical; the
        program of a graph mode is (re)built lazily by the next steps.  The single serial graph (True) is a constructor
        choice only.  bench.py uses this to leave eager mode on a host-constrained node."""
        if self.use_graph or use_graph is True:
            raise ValueError('the single serial graph is chosen at construction; set_launch_mode switches False / '
                             "'forward' / 'segments'")
        if use_graph not in (False, 'forward', 'segments'):
            raise ValueError("set_launch_mode: False, 'forward' or 'segments'")
        segments, forward = use_graph in ('segments', 'forward'), use_graph == 'forward'
        if self._program is not None and (not segments or forward != self.forward_graph):
            self._program = None
        self.segments, self.forward_graph = segments, forward     # (a trainer's first two steps are eager in any mode, _step)

    @property
    def launch_mode(self):
        return True if self.use_graph else ('forward' if self.forward_graph else ('segments' if self.segments else False))

    def _on_bucket(self, _user, first_layer, stream):
        """pdes_bucket_hook: the weight gradients of layers [first_layer, n) are final on the weight-gradient stream"""
        try:
            off = self.model._conv_off[first_layer]
            side, other = self.eng._side_stream(), None
            if stream is not None and stream != side.cuda_stream and stream == self.eng._side_stream('b').cuda_stream:
                side, other = self.eng._side_stream('b'), side   # (the segment program alternates the two weight-gradient streams)
            if self.bucket_stream == 'wgrad_b':
                # the other weight-gradient stream, behind an event recorded where the reduced gradients are final
                tgt = other if other is not None else self.eng._side_stream('b')
                ev = torch.cuda.Event()
                ev.record(side)
                tgt.wait_event(ev)
                side = tgt
            if self._rccl is not None:
                # bucket A behind the early split-K reduce just enqueued on the weight-gradient stream
                self._rccl.all_reduce_sum_(self.gflat.data_ptr() + 4 * off, self.gflat.numel() - off, side.cuda_stream)
                self._bucket_work = True
            else:
                with torch.cuda.stream(side):
                    self._bucket_work = torch.distributed.all_reduce(self.gflat[off:], op=torch.distributed.ReduceOp.SUM,
                                                                     group=self.pg, async_op=True)
            # the main stream waits for bucket A before bucket B is enqueued (_exchange_rest): for the stream handed to the
            # hook pdes_backward2's final join already does, the event makes it hold for every placement
            self._bucket_event = torch.cuda.Event()
            self._bucket_event.record(side)
            self._bucket_off = off
            return 0
        except Exception as e:                                  # never let an exception cross the C ABI
            self._hook_error = e
            return -1

    # ------------------------------------------------------------------------------------------
    def _compute(self):
        """forward + loss + backward on self.x_static -> gradients in self.gflat, terms in self.terms"""
        L, st = self._L, _lib.stream_ptr()
        m = self.model
        assert m._flat is self.flat, 'the model was re-flattened (moved to another device?) after the trainer was built'
        prof = self.host_prof
        t0 = time.perf_counter() if prof is not None else 0.0
        y = self.eng.forward(self.x_static, True, defer_running=True)
        t1 = time.perf_counter() if prof is not None else 0.0
        tail = self._loss(y, st)
        if not self._grad_clean or m._grad_dirty:      # an autograd backward of the same model shares this buffer
            self.gflat.zero_()
            m._grad_dirty = False
        t2 = time.perf_counter() if prof is not None else 0.0
        hook = self._hook if (self.overlap_allreduce and self.bucket_stream != 'main' and not self.use_graph) else None
        self._hook_error = None
        try:
            self.eng.backward(self.grad_y, tail=tail, bucket_hook=hook)
        except RuntimeError:
            if self._hook_error is not None:                  # the all-reduce of bucket A failed inside the callback
                raise self._hook_error
            raise
        if prof is not None:
            t3 = time.perf_counter()
            prof['forward'] = prof.get('forward', 0.0) + (t1 - t0)
            prof['loss'] = prof.get('loss', 0.0) + (t2 - t1)
            prof['backward'] = prof.get('backward', 0.0) + (t3 - t2)
            if 'series' in prof:                              # per-step phases (bench.py: where a host stall lands)
                prof['series'].append((round((t1 - t0) * 1e3, 3), round((t2 - t1) * 1e3, 3), round((t3 - t2) * 1e3, 3)))
        return m

    def _loss(self, y, st):
        """loss + dL/dy of the network output `y` into self.grad_y; returns the `tail` of Engine.backward"""
        # loss_out = NULL: the per-image partials are reduced (and accumulated for the epoch mean) by the
        # end-of-step launch of the backward, together with the BatchNorm bookkeeping
        rc = self._L.pdes_darcy_loss(self.ctx, self.x_static.data_ptr(), y.data_ptr(), self.grad_y.data_ptr(),
                                     self.partials.data_ptr(), None, self.B, self.n, self.n,
                                     1.0, 1.0, self.wb, self.wb, 1 if self.nl else 0, self.nb1, self.nb2, st)
        _lib.check(rc, 'pdes_darcy_loss')
        return self._tail_args()

    def _set_hyper(self, lr):
        self.step_count += 1
        b1, b2 = self.betas
        vals = (lr, b1, b2, self.eps, self.wd, 1.0 - b1 ** self.step_count, math.sqrt(1.0 - b2 ** self.step_count))
        if self.use_graph:
            # the captured graph is followed by an Adam kernel that reads DEVICE memory.  The pinned source is
            # reused every step, so step() waits (one event, graph path only) until the previous copy has read it
            self._hyper_host[:7] = torch.tensor(vals, dtype=torch.float32)
            self.hyper.copy_(self._hyper_host, non_blocking=True)
            self._hyper_event.record()
        else:
            self._hyper_args[:7] = vals          # by value in the kernel arguments: no copy, nothing to race with

    def load_batch(self, data, index):
        """gather minibatch `data[index]` (device tensors) straight into the static input buffer (one launch instead of
        index_select + copy); follow with step(None, lr)"""
        torch.index_select(data, 0, index, out=self.x_static)

    def step(self, x=None, lr=None):
        """one training step on minibatch `x` (device tensor (B,C,H,W); None = reuse x_static)."""
        with _lib.device_guard(self.dev):
            self._step(x, lr)

    def _step(self, x, lr):
        prof = self.host_prof
        ts = time.perf_counter() if prof is not None else 0.0
        if x is not None:
            self.x_static.copy_(x)
        if self.use_graph and self.step_count:
            self._hyper_event.synchronize()                       # previous step's hyper copy has read the pinned buffer
        self._set_hyper(self.lr if lr is None else lr)
        if self.use_graph:
            if self._graph is None:
                self._capture()
            self._graph.replay()
        elif self.segments and self.step_count > 1:
            self._run_segments()
        else:
            self._compute()                                   # (segments: the first step is eager -- lazy initialisations, a clean arena)
        self.n_accum += 1
        if self._hook is not None:                            # data parallel (a group of ONE rank still runs the path)
            self._exchange_rest()
            wd = getattr(self, '_watchdog', None)
            if wd is not None and wd.timeout > 0:             # one event behind the step's last collective (main stream)
                ev = torch.cuda.Event()
                ev.record()
                wd.arm(f'the gradient all-reduce of training step {self.step_count}', ev.query)
        if self.use_graph:
            rc = self._L.pdes_adam_step(self.flat.data_ptr(), self.gflat.data_ptr(), self.exp_avg.data_ptr(),
                                        self.exp_avg_sq.data_ptr(), self.hyper.data_ptr(), 1.0 / self.world,
                                        self.flat.numel(), _lib.stream_ptr())
        else:
            # the kernel clears the gradient buffer after reading it: the next step needs no fill launch
            # ... and the fp64 statistics arena of this step (read last by the end-of-step launch): no fill launch there either
            rc = self._L.pdes_adam_step_host2(self.flat.data_ptr(), self.gflat.data_ptr(), self.exp_avg.data_ptr(),
                                              self.exp_avg_sq.data_ptr(), self._hyper_args, 1.0 / self.world, 1,
                                              self.flat.numel(), self.eng.arena.data_ptr(), self.eng.arena.numel(),
                                              _lib.stream_ptr())
            self._grad_clean = True
            self.eng.arena_clean = True
        _lib.check(rc, 'pdes_adam_step')
        if prof is not None:
            prof['step'] = prof.get('step', 0.0) + (time.perf_counter() - ts)
            prof['n'] = prof.get('n', 0) + 1

    def _run_segments(self):
        """the step as linear hipGraphs (StepProgram): built after the first, eager step"""
        from .models.codec import StepProgram
        m = self.model
        assert m._flat is self.flat, 'the model was re-flattened (moved to another device?) after the trainer was built'
        if self._program is None:
            tail_box = []
            self._program = StepProgram(self.eng, lambda st: tail_box.append(self._loss(self.eng.X['out'], st)),
                                        self._tail_args(), self.grad_y, self.seg_max, forward_only=self.forward_graph,
                                        split_w=os.environ.get('PDES_SEG_SPLITW', '0') == '1')
        if not self._grad_clean or m._grad_dirty:
            self.gflat.zero_()
            m._grad_dirty = False
        if not self.eng.arena_clean:
            self.eng.arena.zero_()
        self.eng.arena_clean = False
        hook = self._hook if (self.overlap_allreduce and self.bucket_stream != 'main') else None
        self._hook_error = None
        try:
            self._program.run(hook)
            if self.forward_graph:                             # the backward pass as eager launches on three streams
                self.eng.backward(self.grad_y, tail=self._tail_args(), bucket_hook=hook)
        except RuntimeError:
            if self._hook_error is not None:
                raise self._hook_error
            raise

    def _tail_args(self):
        return (True, self.partials, self.B, self.n, self.n, 1.0, 1.0, self.wb, self.wb, self.terms, self.terms_accum)

    def _exchange_rest(self):
        """finish the gradient exchange of this step: what the early bucket (if the hook ran) did not cover, then the
        main stream waits for all of it"""
        work, off = self._bucket_work, self._bucket_off
        self._bucket_work = None
        ev, self._bucket_event = self._bucket_event, None
        if work is not None and ev is not None:               # bucket A is complete before anything else of the exchange starts
            torch.cuda.current_stream(self.dev).wait_event(ev)
        if self._rccl is not None:
            n = self.gflat.numel() if work is None else off   # no early bucket: everything; else the head of the buffer
            if n:                                             # on the main stream: behind the end-of-step launch, in front of Adam
                self._rccl.all_reduce_sum_(self.gflat.data_ptr(), n, _lib.stream_ptr(self.dev))
            return
        if work is not None:                                  # bucket A is in flight since the middle of the backward pass
            if off:
                torch.distributed.all_reduce(self.gflat[:off], op=torch.distributed.ReduceOp.SUM, group=self.pg)
            work.wait()                                       # the main stream waits for bucket A
        else:
            torch.distributed.all_reduce(self.gflat, op=torch.distributed.ReduceOp.SUM, group=self.pg)

    def exchange_standalone(self):
        """the whole gradient buffer through the step's exchange path with nothing to overlap (bench.py: what the step
        would pay for the all-reduce if it were not hidden under the backward pass)"""
        self._bucket_work, self._bucket_off = None, 0
        self._exchange_rest()

    def _capture(self):
        # warm the allocator / lazy inits on a side stream, then capture the compute part once
        s = torch.cuda.Stream(device=self.dev)
        s.wait_stream(torch.cuda.current_stream(self.dev))
        saved = self.terms_accum.clone()
        # the warm-up executes one real forward: save / restore what it mutates besides gradients
        # (BatchNorm running statistics and num_batches_tracked, the loss accumulator)
        bns = [m for m in self.model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
        snap = [(m.running_mean.clone(), m.running_var.clone(), m.num_batches_tracked.clone()) for m in bns]
        with torch.cuda.stream(s):
            self._compute()
        torch.cuda.current_stream(self.dev).wait_stream(s)
        torch.cuda.synchronize(self.dev)
        for m, (rm, rv, nb) in zip(bns, snap):
            m.running_mean.copy_(rm)
            m.running_var.copy_(rv)
            m.num_batches_tracked.copy_(nb)
        self.terms_accum.copy_(saved)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._compute()
        self.terms_accum.copy_(saved)
        self._graph = g

    def epoch_means(self):
        """mean of {loss, const, cont, dirichlet, neumann} since the last call (ONE host sync)."""
        t = (self.terms_accum / max(self.n_accum, 1)).cpu().tolist()
        self.terms_accum.zero_()
        self.n_accum = 0
        return t


class MaxLikelihoodTrainer(MixedResidualTrainer):
    """The data-driven loop body of train_codec_max_likelihood.py:197-211 (same DenseED, F.mse_loss against FEniCS
    targets) on the same fused step: only the loss launch differs (pdes_mse_loss instead of pdes_darcy_loss)."""

    def __init__(self, model, batch_size, imsize=64, out_channels=3, **kw):
        super().__init__(model, batch_size, imsize, **kw)
        self.target_static = torch.zeros((batch_size, out_channels, imsize, imsize), device=self.dev)
        n = self.target_static.numel()
        self._mse_partials = torch.empty(self._L.pdes_mse_partials(n), device=self.dev, dtype=torch.float64)

    def _loss(self, y, st):
        # the finalize launch adds the batch loss to terms_accum[0] (epoch mean, one host sync per epoch)
        rc = self._L.pdes_mse_loss(y.data_ptr(), self.target_static.data_ptr(), self.grad_y.data_ptr(),
                                   self._mse_partials.data_ptr(), self.terms.data_ptr(), self.terms_accum.data_ptr(),
                                   y.numel(), st)
        _lib.check(rc, 'pdes_mse_loss')
        return self._tail_args()

    def _tail_args(self):
        return (True, None, self.B, self.n, self.n, 0.0, 0.0, 0.0, 0.0, None, None)

    def step(self, x=None, target=None, lr=None):
        """one step on (input, target); None = reuse the static buffers"""
        if target is not None:
            self.target_static.copy_(target)
        super().step(x, lr)


class ReverseKLTrainer:
    """The loop body of train_cglow_reverse_kl.py:245-272 as one fused step: draw the latents' noise, generate
    y ~ p(y|x) with log p(y|x) (the conditional Glow's descriptor chain), loss = beta * loss_pde(y; x) + E[log p] / ln 2 /
    n_pixels with the fused Sobel + Darcy-residual kernel, backward over the same chain, flat all-reduce (one process per
    GPU), one-cycle learning rate, flat Adam -- no autograd graph, no per-step host synchronisation.  The loss terms are
    accumulated on the device and read once per epoch."""

    def __init__(self, model, batch_size, imsize=32, lr=1.5e-3, weight_decay=0.0, weight_bound=50.0, beta=150.0,
                 betas=(0.9, 0.999), eps=1e-8, device=None, process_group=None):
        self.model = model
        self.B, self.n = batch_size, imsize
        self.dev = torch.device(device if device is not None else 'cuda:0')
        if self.dev.type != 'cuda':
            raise RuntimeError('ReverseKLTrainer runs on an MI355X only (no CPU fallback)')
        self.host_threads = parallel.limit_host_threads(local_world=parallel.local_world_size(1))
        self.wb, self.beta = float(weight_bound), float(beta)
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.pg = process_group
        self.world = 1
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
        self.dp = self.world > 1 or process_group is not None
        model.to(self.dev)
        probe = torch.zeros((batch_size, 1, imsize, imsize), device=self.dev)
        with _lib.device_guard(self.dev):
            self.eng = model._engine(probe)
        self.eng.reserved = True
        self.ctx = self.eng.ctx
        self.x_static = self.eng.X['in']
        self.x_static.zero_()
        self.flat, self.gflat = model._flat, model._gscratch
        self._rccl = None
        if self.dp:
            parallel.broadcast_parameters(self.flat, process_group)
            self._rccl = parallel.make_direct_rccl(process_group, self.dev)      # ncclAllReduce by pointer (or None: torch's)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self._hyper_args = (ctypes.c_float * 8)()
        self.step_count = 0
        C = model.y_channels
        self.grad_y = torch.empty((batch_size, C, imsize, imsize), device=self.dev)
        self.partials = torch.empty((_lib.loss_partial_rows(batch_size, imsize, imsize, 0), 4), device=self.dev)
        self.terms = torch.zeros(5, device=self.dev)              # {beta * loss_pde, constitutive, continuity, dirichlet, neumann}
        self.terms_accum = torch.zeros(6, device=self.dev, dtype=torch.float64)     # ... + sum_b log p(y_b|x_b)
        self.n_accum = 0
        self._ne_scale = 1.0 / (batch_size * math.log(2.0) * C * imsize * imsize)
        self.eng.glogp.fill_(self._ne_scale)                      # d(neg_entropy)/d(logp_b): the same for every sample
        self._eps_bufs = [self.eng.X[name] for _, name in sorted(model._meta['eps'].items())]
        self._grad_clean = False
        self._L = _lib.lib()

    close = MixedResidualTrainer.close
    __del__ = MixedResidualTrainer.__del__

    def load_batch(self, data, index):
        torch.index_select(data, 0, index, out=self.x_static)

    def step(self, x=None, lr=None, eps_list=None):
        """one step on minibatch `x` (None = reuse the static input buffer); eps_list (tests): the latents' noise"""
        with _lib.device_guard(self.dev):
            self._step(x, lr, eps_list)

    def _step(self, x, lr, eps_list):
        L, st = self._L, _lib.stream_ptr()
        m, eng = self.model, self.eng
        assert m._flat is self.flat, 'the model was re-flattened after the trainer was built'
        if x is not None:
            self.x_static.copy_(x)
        for k, buf in enumerate(self._eps_bufs):
            if eps_list is None:
                buf.normal_()
            else:
                buf.copy_(eps_list[k])
        eng.run({}, True)
        y = eng.X['out']
        b = self.beta
        rc = L.pdes_darcy_loss(self.ctx, self.x_static.data_ptr(), y.data_ptr(), self.grad_y.data_ptr(),
                               self.partials.data_ptr(), self.terms.data_ptr(), self.B, self.n, self.n, b, b, b * self.wb,
                               b * self.wb, 0, 0.0, 0.0, st)
        _lib.check(rc, 'pdes_darcy_loss')
        self.terms_accum[:5] += self.terms
        self.terms_accum[5] += eng.logp.sum()
        self.n_accum += 1
        if not self._grad_clean or m._grad_dirty:
            self.gflat.zero_()
            m._grad_dirty = False
        eng.backward(self.grad_y, eng.glogp)
        if self._rccl is not None:
            self._rccl.all_reduce_sum_(self.gflat.data_ptr(), self.gflat.numel(), st)
        elif self.dp:
            torch.distributed.all_reduce(self.gflat, op=torch.distributed.ReduceOp.SUM, group=self.pg)
        self.step_count += 1
        b1, b2 = self.betas
        self._hyper_args[:7] = (self.lr if lr is None else lr, b1, b2, self.eps, self.wd, 1.0 - b1 ** self.step_count,
                                math.sqrt(1.0 - b2 ** self.step_count))
        rc = L.pdes_adam_step_host(self.flat.data_ptr(), self.gflat.data_ptr(), self.exp_avg.data_ptr(),
                                   self.exp_avg_sq.data_ptr(), self._hyper_args, 1.0 / self.world, 1, self.flat.numel(), st)
        _lib.check(rc, 'pdes_adam_step_host')
        self._grad_clean = True

    def epoch_means(self):
        """means since the last call of {loss, residual (constitutive + continuity), boundary (dirichlet + neumann),
        neg_entropy} (ONE host sync)"""
        t = (self.terms_accum / max(self.n_accum, 1)).cpu().tolist()
        self.terms_accum.zero_()
        self.n_accum = 0
        neg_entropy = t[5] * self._ne_scale
        return [t[0] + neg_entropy, t[1] + t[2], t[3] + t[4], neg_entropy]
