"""world_size-2 rehearsal (gloo, CPU) of the parameter-shift shard: each rank evaluates its (interleaved) share of
the 2P shifted circuits (here with the oracle standing in for the GPU engine), one all-gather of the
per-parameter gradient scalars, identical optimiser step on every rank."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import circuit as oc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world_size, port, P, ansatz, n, L, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        from tensornetworks_amd.paramshift_shard import shard_params, all_gather_grad, world
        assert world() == (rank, world_size)
        rng = np.random.default_rng(0)                      # replicated theta / dLdq
        theta = rng.uniform(-1, 1, P)
        dLdq = rng.normal(size=2 ** n)
        mine = range(*shard_params(P, rank, world_size))        # rank, rank + W, ...
        local = np.zeros(len(mine))
        for i, p in enumerate(mine):
            tp = theta.copy(); tp[p] += np.pi / 2
            tm = theta.copy(); tm[p] -= np.pi / 2
            local[i] = 0.5 * dLdq @ (oc.probs(ansatz, n, L, tp) - oc.probs(ansatz, n, L, tm))
        full = all_gather_grad(torch.as_tensor(local), P)
        assert full.shape == (P,)
        th = torch.nn.Parameter(torch.as_tensor(theta, dtype=torch.float32))
        opt = torch.optim.Adam([th], lr=0.01)
        th.grad = full.to(torch.float32)
        torch.nn.utils.clip_grad_norm_([th], 10.0)
        opt.step()
        np.save(os.path.join(out_dir, f"grad_{rank}.npy"), full.numpy())
        np.save(os.path.join(out_dir, f"theta_{rank}.npy"), th.detach().numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world_size,n,L", [(2, 3, 2), (3, 4, 1)])
def test_sharded_gradient_equals_full(tmp_path, world_size, n, L):
    ansatz = "hardware_efficient"
    P = oc.num_params(ansatz, n, L)
    mp.spawn(_worker, args=(world_size, _free_port(), P, ansatz, n, L, str(tmp_path)), nprocs=world_size, join=True)
    rng = np.random.default_rng(0)
    theta = rng.uniform(-1, 1, P)
    dLdq = rng.normal(size=2 ** n)
    ref = oc.paramshift_vjp(ansatz, n, L, theta, dLdq)
    grads = [np.load(tmp_path / f"grad_{r}.npy") for r in range(world_size)]
    thetas = [np.load(tmp_path / f"theta_{r}.npy") for r in range(world_size)]
    for r in range(world_size):
        np.testing.assert_allclose(grads[r], ref, rtol=1e-12, atol=1e-14)
        np.testing.assert_array_equal(grads[r], grads[0])          # bitwise identical on every rank
        np.testing.assert_array_equal(thetas[r], thetas[0])        # => identical update, no broadcast needed


def _sym_worker(rank, world_size, port, n, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        from tensornetworks_amd import backend
        from tensornetworks_amd.paramshift_shard import all_reduce_sum
        from tensornetworks_amd._ext import lib
        from oracle import stein as os_
        R = lib().bornvi_stein_sym_strip_rows()
        N = 2 ** n
        rng = np.random.default_rng(5)
        S = rng.normal(size=(N, n))
        K = os_.gram_closed_form(S, n, 1.0)
        q = rng.random(N); q /= q.sum()
        (pa, pb), (l0, l1), (h0, h1) = backend.sym_pair_shard(n, rank, world_size)
        assert (l1 - l0) == (pb - pa) * R == (h1 - h0)
        msg = np.zeros(N + 1)
        ns = N // R
        for s in list(range(pa, pb)) + list(range(ns - pb, ns - pa)):      # what quadform_sym_kernel does per strip
            i0 = s * R
            blk = K[i0:i0 + R]
            msg[i0:i0 + R] += blk[:, i0:] @ q[i0:]                            # row part: columns >= i0
            msg[i0 + R:N] += q[i0:i0 + R] @ blk[:, i0 + R:]                   # column part: columns right of the diagonal block
        msg[N] = q @ msg[:N]
        t = torch.as_tensor(msg)
        all_reduce_sum(t)
        np.save(os.path.join(out_dir, f"sym_{rank}.npy"), t.numpy())
        if rank == 0:
            np.save(os.path.join(out_dir, "sym_ref.npy"), np.concatenate([K @ q, [q @ K @ q]]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world_size,n", [(2, 9), (3, 10)])
def test_strip_pair_shard_all_reduce(tmp_path, world_size, n):
    """The strip pairs dealt to the ranks cover the upper triangle exactly once: the all-reduced shares are K q and
    q^T K q, bit-identical on every rank (the partition and the collective of the multi-GPU contraction)."""
    mp.spawn(_sym_worker, args=(world_size, _free_port(), n, str(tmp_path)), nprocs=world_size, join=True)
    ref = np.load(tmp_path / "sym_ref.npy")
    outs = [np.load(tmp_path / f"sym_{r}.npy") for r in range(world_size)]
    for r in range(world_size):
        np.testing.assert_allclose(outs[r], ref, rtol=1e-12, atol=1e-14 * np.abs(ref).max())
        np.testing.assert_array_equal(outs[r], outs[0])
