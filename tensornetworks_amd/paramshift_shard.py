"""Sharding of the 2P parameter-shift circuit evaluations across the GPUs of one node.

The shifted circuits of a training step are independent given theta and dL/dq, so rank r of W
evaluates the parameters r, r + W, r + 2W, ... (an interleaved deal: with prefix sharing a parameter
of an early layer costs more passes than one of a late layer, and parameters are numbered layer by
layer -- contiguous slices would leave rank 0 with all the expensive ones) and the per-parameter
gradient scalars (8 bytes each) are exchanged with ONE all-gather per step (RCCL over xGMI when the process group uses the
'nccl' backend; 'gloo' in the CPU tests).  theta, optimiser state, S and K_p are replicated, and
every rank applies the identical update, so no broadcast is needed.

The reference has no distributed code; this is a new design.  Note that
what is gathered are GRADIENT scalars 1/2 dLdq.(q+ - q-), not shifted KSD values: the KSD is
quadratic in q, so the two-term shift rule does not apply to it directly.
"""
import torch
import torch.distributed as dist


SOLO = "solo"     # process_group value: never shard, even inside an initialised torch.distributed job


def world(group=None):
    """(rank, world_size) of `group`, or (0, 1) when torch.distributed is not initialised (or group is SOLO)."""
    if isinstance(group, str) and group == SOLO:
        return 0, 1
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_range(num_params, rank, world_size):
    """Contiguous slice of parameters owned by `rank`; equal-sized chunks of ceil(P/W) (the last
    ranks may own fewer, or none)."""
    chunk = -(-num_params // world_size)
    lo = min(num_params, rank * chunk)
    hi = min(num_params, lo + chunk)
    return lo, hi


def shard_params(num_params, rank, world_size):
    """(first, stop, stride) of the parameters owned by `rank`: range(first, stop, stride) = rank, rank + W, ...
    (at most ceil(P/W) of them; the last ranks may own one fewer, or none)."""
    return min(rank, num_params), num_params, world_size


def all_gather_flat(out, msg, group=None):
    """all_gather_into_tensor for equal-sized float64 messages.  With the 'nccl' backend (RCCL) the device
    tensors go straight onto xGMI; a 'gloo' group (CPU rehearsal, or several ranks sharing one GPU) stages
    the few KB through host memory."""
    backend = dist.get_backend(group)
    if backend == "gloo" and msg.is_cuda:
        out_h = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(out_h, msg.cpu(), group=group)
        out.copy_(out_h)
    else:
        dist.all_gather_into_tensor(out, msg, group=group)
    return out


def all_reduce_sum(msg, group=None):
    """In-place sum over the ranks of a float64 vector (the partial K q of the strip-pair shard); every rank ends
    with the same bits.  'gloo' groups with device tensors stage through host memory like all_gather_flat."""
    backend = dist.get_backend(group)
    if backend == "gloo" and msg.is_cuda:
        host = msg.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        msg.copy_(host)
    else:
        dist.all_reduce(msg, op=dist.ReduceOp.SUM, group=group)
    return msg


def all_gather_grad(local_grad, num_params, group=None):
    """local_grad: this rank's gradient scalars for the parameters of shard_params (float64, any device the backend
    supports) -> full [P] vector in parameter order, identical on every rank."""
    rank, ws = world(group)
    if ws == 1:
        return local_grad
    chunk = -(-num_params // ws)
    padded = torch.zeros(chunk, dtype=local_grad.dtype, device=local_grad.device)
    padded[: local_grad.numel()] = local_grad
    full = torch.empty(chunk * ws, dtype=local_grad.dtype, device=local_grad.device)
    all_gather_flat(full, padded, group)
    # full[r * chunk + i] belongs to parameter i * W + r
    return full.view(ws, chunk).t().reshape(-1)[:num_params].contiguous()
