"""Child-process bodies of the multi-rank GPU tests (not a test module).

The children are forked from a multiprocessing *forkserver* that conftest.py starts before anything in the pytest
process has touched the GPU: a fresh process initialises HIP itself, and no process that has initialised the GPU is
ever replaced by exec (which the GPU pool forbids)."""
import os
import sys
import traceback

import numpy as np


def _init(rank, world_size, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"] = str(rank)
    os.environ["LOCAL_RANK"] = "0"            # every rank shares cuda:0 of the one-GPU box
    os.environ["WORLD_SIZE"] = str(world_size)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def trainer_rank(rank, world_size, port, n, L, symmetric, out_dir):
    """One rank of a W-rank KSD-gradient step: a real torch.distributed group (gloo; the ranks share one GPU), the
    product trainer end to end -- strip-pair or row shard of K_p, all-reduce / all-gather of the contraction,
    interleaved deal of the 2P shifted circuits, all-gather of the gradient scalars, one optimiser step."""
    try:
        _init(rank, world_size, port)
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world_size)
        try:
            from tensornetworks_amd.bayesian_network import synthetic_network
            from tensornetworks_amd.ksd_vi_quantum import KSDVariationalInference
            torch.cuda.set_device(0)
            bn, lat, obs, x = synthetic_network(n, seed=1)
            torch.manual_seed(7)
            vi = KSDVariationalInference(bn, lat, obs, qbm_num_latent_vars=n, qbm_ansatz_layers=L,
                                         pytorch_device="cuda:0", gram_mode="dense")
            vi.symmetric_contraction = bool(symmetric)
            vi._prepare_stein(x)
            if symmetric:
                assert vi._K_pairs is not None and vi._K.shape[0] == (1 << n) // world_size
            else:
                assert vi._K_rows is not None and vi._K.shape[0] <= -(-(1 << n) // world_size)
            loss, grad, q = vi.ksd_and_grad()
            params, opt, sched = vi.make_optimizer(0.01, 3, True, "adam", (0.9, 0.999))
            l2, gn, _ = vi.training_step(params, opt, sched, 10.0)
            torch.cuda.synchronize()
            np.savez(os.path.join(out_dir, f"rank{rank}.npz"), loss=loss.cpu().numpy(), grad=grad.cpu().numpy(),
                     q=q.cpu().numpy(), theta=vi.born_machine.theta.detach().cpu().numpy(), loss2=np.float64(l2),
                     gn=np.float64(float(gn)))
            dist.barrier()
        finally:
            dist.destroy_process_group()
    except BaseException:
        with open(os.path.join(out_dir, f"rank{rank}.err"), "w") as f:
            traceback.print_exc(file=f)
        raise


def bench_rank(rank, world_size, port, argv, out_dir):
    """One rank of `bench.py --gpus W ...` exactly as torchrun would start it (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in the environment), with the gloo backend because the ranks share one GPU here."""
    try:
        _init(rank, world_size, port)
        os.environ["BORNVI_DIST_BACKEND"] = "gloo"
        repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        if repo not in sys.path:
            sys.path.insert(0, repo)
        import contextlib
        import io
        import bench
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            bench.main(list(argv))
        with open(os.path.join(out_dir, f"bench{rank}.out"), "w") as f:
            f.write(buf.getvalue())
    except BaseException:
        with open(os.path.join(out_dir, f"bench{rank}.err"), "w") as f:
            traceback.print_exc(file=f)
        raise


def rccl_single_rank(rank, world_size, port, out_dir):
    """A ONE-rank 'nccl' (= RCCL) group on cuda:0: the collective wrappers of paramshift_shard and the calls bench.py's
    Dist makes, on device tensors, through RCCL itself (two ranks cannot share a GPU under RCCL, so the one-GPU box can
    only check that the library initialises with `device_id=`, accepts float64 device tensors and returns the right
    layout; the W > 1 arithmetic is what the gloo tests pin)."""
    try:
        _init(rank, world_size, port)
        import torch
        import torch.distributed as dist
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=dev)
        try:
            from tensornetworks_amd import paramshift_shard as shard
            assert dist.get_backend() == "nccl" and shard.world() == (0, 1)
            g = torch.Generator(device="cpu").manual_seed(3)
            msg = torch.randn(4097, dtype=torch.float64, generator=g).to(dev)
            out = torch.empty_like(msg)
            shard.all_gather_flat(out, msg)
            red = msg.clone()
            shard.all_reduce_sum(red)
            t = torch.tensor([1.25], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            objs = [None]
            dist.all_gather_object(objs, {"circuits": 1.0})
            dist.barrier()
            torch.cuda.synchronize()
            ok = bool(torch.equal(out, msg) and torch.equal(red, msg) and float(t.item()) == 1.25 and objs == [{"circuits": 1.0}])
            np.savez(os.path.join(out_dir, "rccl.npz"), ok=np.bool_(ok))
        finally:
            dist.destroy_process_group()
    except BaseException:
        with open(os.path.join(out_dir, f"rank{rank}.err"), "w") as f:
            traceback.print_exc(file=f)
        raise


def cold_batch(rank, world_size, port, n, L, read_map, out_dir, reg_wires=3):
    """A FRESH process runs one full parameter-shift batch and checks every row's sum: the first launches of a process
    find cold instruction caches and TLBs, the waves of a workgroup drift apart, and a missing barrier shows (the
    cross-group read map of round 2 produced wrong rows only here)."""
    try:
        import torch
        from tensornetworks_amd import backend as be
        dev = torch.device("cuda", 0)
        be.set_option(dev, "reg_wires", int(reg_wires))
        be.set_option(dev, "read_map", int(read_map))
        P = be.num_params("hardware_efficient", n, L)
        g = torch.Generator().manual_seed(0)
        th = (0.1 * torch.randn(P, generator=g, dtype=torch.float32)).double().to(dev)
        full = be.paramshift_probs("hardware_efficient", n, L, th, 0, P, include_base=True)
        worst = float((full.sum(dim=1) - 1.0).abs().max())
        # the base row and six shifted rows go back to the parent, which compares them with the oracle's C port (a
        # normalised wrong row would pass the sum check); a digest of every row lets two cold runs be compared bitwise
        picks = [0, P // 2 + 7, P - 1]
        rows = [0] + [r for p_ in picks for r in (1 + 2 * p_, 2 + 2 * p_)]
        import hashlib
        digest = hashlib.sha256(full.cpu().numpy().tobytes()).hexdigest()
        np.savez(os.path.join(out_dir, f"cold{rank}.npz"), worst=np.float64(worst), rows=full[rows].cpu().numpy(),
                 theta=th.cpu().numpy(), picks=np.array(picks), digest=np.array(digest))
    except BaseException:
        with open(os.path.join(out_dir, f"rank{rank}.err"), "w") as f:
            traceback.print_exc(file=f)
        raise
