"""Batch-parallel execution across the GPUs of one node.

The reference has no distributed code (main.py:82 picks one device).  Fixed-grid trajectories are independent per
sample (no BatchNorm in f; GroupNorm in the ConvGRU cell is per sample), so the forward needs NO collective: every
rank integrates a contiguous slice of the batch with replicated weights.  Training adds exactly one collective per
step: a single flattened-bucket all-reduce (sum, then / world: the loss is a batch mean, models/ODEConvGRU.py:96-98)
of the parameter gradients -- 0.74 MB for the dynamics, 4.2 MB for the whole ODEConvGRU, i.e. latency-bound on xGMI,
so it is issued ONCE after backward rather than bucketed and overlapped.
Backend: torch.distributed "nccl" (= RCCL) on GPUs, "gloo" in the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_bounds(n, rank, world):
    """Contiguous, balanced slice [lo, hi) of n items for `rank` (the first n % world ranks get one extra)."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(x, rank=None, world=None, dim=0):
    """This rank's slice of a batch-first tensor (no copy for dim 0)."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lo, hi = shard_bounds(x.shape[dim], rank, world)
    return x.narrow(dim, lo, hi - lo)


def allreduce_gradients(params, group=None, average=True):
    """ONE all-reduce over one flattened bucket of every .grad (missing grads count as zeros so that every rank sends
    the same layout).  Returns the number of elements reduced."""
    params = [p for p in params if p.requires_grad]
    if not params:
        return 0
    world = dist.get_world_size(group)
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat.div_(world)
    off = 0
    for p in params:
        n = p.numel()
        g = flat[off:off + n].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n
    return flat.numel()


def broadcast_parameters(module, src=0, group=None):
    """Make every rank start from rank `src`'s weights (one flattened broadcast)."""
    ps = [p for p in module.parameters()]
    if not ps:
        return
    flat = torch.cat([p.detach().reshape(-1) for p in ps])
    dist.broadcast(flat, src=src, group=group)
    off = 0
    with torch.no_grad():
        for p in ps:
            n = p.numel()
            p.copy_(flat[off:off + n].view_as(p))
            off += n


def gather_batch(x_local, group=None):
    """All-gather variable-size batch shards back into the full batch (evaluation / tests)."""
    world = dist.get_world_size(group)
    sizes = [torch.zeros(1, dtype=torch.long, device=x_local.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([x_local.shape[0]], dtype=torch.long, device=x_local.device), group=group)
    mx = int(max(int(s) for s in sizes))
    pad = torch.zeros((mx,) + tuple(x_local.shape[1:]), dtype=x_local.dtype, device=x_local.device)
    pad[:x_local.shape[0]] = x_local
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad, group=group)
    return torch.cat([o[:int(s)] for o, s in zip(outs, sizes)], 0)


_global_ctl = None   # (callback object, scratch tensor, group): kept alive while installed


def enable_global_step_control(device, group=None):
    """dopri5 under batch sharding with torchdiffeq's semantics: ONE error norm over the whole (global) batch.

    Installs an all-reduce hook in the HIP library (include/odecgru_hip.h: odehip_set_norm_allreduce): every sum of squares
    the step controller uses is summed over the ranks of `group` first (a 4..8-byte RCCL all-reduce per attempted step), so
    all ranks take the accept/reject decisions a single device holding the full batch would take.  Every rank must call
    odeint the same number of times with the same t.  Without it each rank adapts on its own shard (agrees to O(rtol))."""
    global _global_ctl
    import ctypes
    from . import _lib
    scratch = torch.zeros(8, dtype=torch.float32, device=device)
    world = dist.get_world_size(group)

    def _cb(ptr, n, stream, user):
        try:
            dist.all_reduce(scratch[:n], op=dist.ReduceOp.SUM, group=group)
            return 0
        except Exception:   # must not propagate through the C frame
            import traceback
            traceback.print_exc()
            return 1
    cb = _lib.ALLREDUCE_FN(_cb)
    _lib.check(_lib.load().odehip_set_norm_allreduce(ctypes.cast(cb, ctypes.c_void_p), None, world, scratch.data_ptr()))
    _global_ctl = (cb, scratch, group)


def disable_global_step_control():
    global _global_ctl
    from . import _lib
    _lib.check(_lib.load().odehip_set_norm_allreduce(None, None, 1, None))
    _global_ctl = None
