"""Host-side data-parallel logic of the mixed-residual trainer (one process per GPU).

The reference has no distributed code at all; data parallelism is what this build adds:
  * every rank holds the dataset replica and the same permutation; rank r takes samples
    [step*GB + r*B, step*GB + (r+1)*B) of it (`shard_indices`);
  * BatchNorm statistics stay rank-local (what DistributedDataParallel does);
  * gradients are ONE flat fp32 buffer -> one all-reduce(SUM) per step over RCCL/xGMI (2.96 MB for the
    default net: latency-bound), then Adam with grad_scale = 1/world_size on every rank;
  * parameters start identical (same seed, checked/enforced by `broadcast_parameters`).
These helpers are pure torch (`torch.distributed` with the `nccl` backend = RCCL on ROCm; the CPU
tests run them over `gloo` with world_size 2).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """torchrun-style rendezvous: returns (rank, local_rank, world_size); world 1 = no process group"""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        kw = {}
        if backend == 'nccl':
            torch.cuda.set_device(local)
            kw['device_id'] = torch.device('cuda', local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local, world


def shard_indices(perm, step, batch_size, rank, world_size):
    """this rank's minibatch indices of global step `step` (contiguous split of the global batch)"""
    gb = batch_size * world_size
    lo = step * gb + rank * batch_size
    return perm[lo:lo + batch_size]


def allreduce_sum_(flat, group=None):
    """in-place SUM all-reduce of the flat gradient buffer (no-op without a process group)"""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def broadcast_parameters(flat, group=None, src=0):
    """make every rank start from rank `src`'s parameters (belt and braces: seeds already agree)"""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)
    return flat


def broadcast_buffers(model, group=None, src=0):
    """BatchNorm running statistics / num_batches_tracked of rank `src` to everyone (after loading a checkpoint;
    during training they stay rank-local and rank 0's are what a checkpoint stores)"""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        for b in model.buffers():
            dist.broadcast(b, src=src, group=group)
    return model


def mean_over_ranks(values, group=None):
    """list of python floats -> their mean over the ranks (one small all-reduce; identity without a group).
    Used once per epoch for the logged loss terms: every rank holds the mean over ITS shards."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return list(values)
    dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(group) == 'nccl' else torch.device('cpu')
    t = torch.tensor(list(values), dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return (t / dist.get_world_size(group)).cpu().tolist()


def adam_reference_(param, grad, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8,
                    weight_decay=0.0, grad_scale=1.0):
    """torch restatement of the flat HIP Adam kernel (`pdes_adam_step`): used by the CPU tests of the
    data-parallel path and by the GPU test of the kernel; torch.optim.Adam semantics."""
    g = grad * grad_scale
    if weight_decay != 0.0:
        g = g + weight_decay * param
    b1, b2 = betas
    exp_avg.lerp_(g, 1 - b1)
    exp_avg_sq.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2_sqrt = (1 - b2 ** step) ** 0.5
    denom = (exp_avg_sq.sqrt() / bc2_sqrt).add_(eps)
    param.addcdiv_(exp_avg, denom, value=-lr / bc1)
    return param
