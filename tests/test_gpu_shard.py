"""Multi-rank paths of the PRODUCT trainer on the GPU: real torch.distributed process groups (gloo, ranks sharing
cuda:0 of the one-GPU box), KSDVariationalInference end to end.  The 8-GPU RCCL run itself belongs to the driver; what
is pinned here is that the sharded step computes the single-process step (reference epoch body
ksd_vi_quantum.py:110-161; the reference itself is single-process)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import run_ranks
import shard_worker

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda", 0)


def _solo_step(n, L, symmetric):
    from tensornetworks_amd.bayesian_network import synthetic_network
    from tensornetworks_amd.ksd_vi_quantum import KSDVariationalInference
    bn, lat, obs, x = synthetic_network(n, seed=1)
    torch.manual_seed(7)
    vi = KSDVariationalInference(bn, lat, obs, qbm_num_latent_vars=n, qbm_ansatz_layers=L, pytorch_device="cuda:0",
                                 gram_mode="dense")
    vi.symmetric_contraction = bool(symmetric)
    vi._prepare_stein(x)
    loss, grad, q = vi.ksd_and_grad()
    params, opt, sched = vi.make_optimizer(0.01, 3, True, "adam", (0.9, 0.999))
    l2, gn, _ = vi.training_step(params, opt, sched, 10.0)
    return (loss.cpu().numpy(), grad.cpu().numpy(), q.cpu().numpy(), vi.born_machine.theta.detach().cpu().numpy(), l2,
            float(gn))


def _errors(tmp_path):
    return "\n".join(open(tmp_path / f).read() for f in sorted(os.listdir(tmp_path)) if f.endswith(".err"))


@pytest.mark.parametrize("world_size,n,L,symmetric", [(2, 12, 2, True), (3, 10, 2, False), (4, 13, 1, True)])
def test_sharded_trainer_step_equals_single_process(dev, tmp_path, world_size, n, L, symmetric):
    """W ranks run KSDVariationalInference.ksd_and_grad + one training_step under a real process group: strip-pair
    shard (+ all-reduce of 2^n + 1 doubles) or row shard (+ all-gather; W = 3 does not divide 2^n: ragged last block),
    interleaved parameter deal, gradient all-gather.  Every rank ends with the SAME bits, and they equal the
    single-process step to rounding (the strip sums are associated differently: 1e-12)."""
    codes = run_ranks(shard_worker.trainer_rank, world_size, (n, L, symmetric, str(tmp_path)))
    assert codes == [0] * world_size, (codes, _errors(tmp_path))
    outs = [np.load(tmp_path / f"rank{r}.npz") for r in range(world_size)]
    for r in range(1, world_size):
        for key in ("loss", "grad", "q", "theta", "loss2", "gn"):
            np.testing.assert_array_equal(outs[r][key], outs[0][key], err_msg=f"rank {r} {key}")
    loss, grad, q, theta, l2, gn = _solo_step(n, L, symmetric)
    np.testing.assert_array_equal(outs[0]["q"], q)                       # the base circuit is replicated: same bits
    np.testing.assert_allclose(outs[0]["loss"], loss, rtol=1e-12)
    np.testing.assert_allclose(outs[0]["grad"], grad, rtol=1e-10, atol=1e-12 * np.abs(grad).max())
    np.testing.assert_allclose(outs[0]["theta"], theta, rtol=0, atol=1e-7)
    assert abs(float(outs[0]["loss2"]) - l2) <= 1e-10 * abs(l2) and abs(float(outs[0]["gn"]) - gn) <= 1e-5 * gn


@pytest.mark.parametrize("W,n", [(2, 9), (3, 8), (5, 10)])
def test_row_shard_messages_assemble_to_the_full_contraction(dev, W, n):
    """The row-shard branch of _stein_contract on one GPU: every rank's stein_quadform_rows message (its rows of K q
    followed by its partial of q.y, padded to ceil(N / W) + 1) assembled exactly as the all-gather would deliver them
    equals the one-rank K q and q^T K q."""
    from tensornetworks_amd import backend
    from tensornetworks_amd.paramshift_shard import shard_range
    N = 1 << n
    rng = np.random.default_rng(n + W)
    S = torch.as_tensor(rng.normal(size=(N, n)), device=dev)
    q = torch.as_tensor(rng.random(N), device=dev); q /= q.sum()
    K = backend.stein_gram(S, n, 1.0)
    k2_ref, Y = backend.stein_quadform(K, q, n, want_y=True)
    chunk = -(-N // W)
    full = torch.zeros((W, chunk + 1), dtype=torch.float64, device=dev)
    for r in range(W):
        r0, r1 = shard_range(N, r, W)
        Kr = backend.stein_gram(S, n, 1.0, rows=(r0, r1))
        assert torch.equal(Kr, K[r0:r1])
        part = backend.stein_quadform_rows(Kr, r0, r1, q, n)
        full[r, : r1 - r0] = part[:-1]
        full[r, chunk] = part[-1]
    y = full[:, :chunk].reshape(-1)[:N]
    ksd2 = full[:, chunk].sum()
    assert torch.equal(y, Y[0])
    assert abs(ksd2.item() - k2_ref.item()) <= 1e-13 * abs(k2_ref.item())


def test_bench_dist_selftest_two_ranks(dev, tmp_path):
    """`bench.py --gpus 2 --dist-selftest` as the driver's torchrun would start it (env-variable rendezvous), with gloo
    because both ranks share this box's one GPU: the self-test compares the all-reduced contraction and the gathered
    gradient with an un-sharded recomputation on every rank, then the bench emits its JSON line with per-phase
    milliseconds per rank."""
    argv = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "n12_L4_dense", "--dist-selftest",
            "--no-cpu-baseline", "--no-gate-bench", "--no-extras", "--series", "n8_L4_dense", "--repeats", "1"]
    codes = run_ranks(shard_worker.bench_rank, 2, (argv, str(tmp_path)), timeout=900)
    assert codes == [0, 0], (codes, _errors(tmp_path))
    line = [l for l in open(tmp_path / "bench0.out").read().splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["config"]["workload"] == "n12_L4_dense"
    st = rec["dist_selftest"]
    assert st["ok"] and st["ranks"] == 2 and st["max_rel_err_grad"] < 1e-10 and st["max_rel_err_y"] < 1e-12
    assert len(rec["phase_ms_per_rank"]) == 2 and all("circuits" in p and "allgather" in p for p in rec["phase_ms_per_rank"])
    ser = rec["series"][0]                                # the second workload measured beside the headline, same ranks
    assert ser["workload"] == "n8_L4_dense" and ser["n_gpus"] == 2 and ser["value"] > 0 and len(ser["phase_ms_per_rank"]) == 2
    assert not [l for l in open(tmp_path / "bench1.out").read().splitlines() if l.startswith("{")]   # rank 0 prints


def test_bench_starts_its_own_ranks(dev, tmp_path):
    """`python bench.py --gpus 2` with NO rendezvous in the environment (the way the driver starts the --gpus 1 run): the
    process starts the two ranks itself before touching the GPU (a child `python -m torch.distributed.run`), relays rank
    0's JSON line and the children's exit code.  gloo here because the ranks share this box's one GPU (bench.py picks
    it when there are fewer GPUs than ranks); the sharded-vs-unsharded self-test is on by default for N > 1."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                             "BORNVI_DIST_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--workload", "n12_L4_dense", "--no-cpu-baseline", "--no-gate-bench", "--no-extras", "--series", "none",
                        "--repeats", "1"], env=env, cwd=repo, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["dist_backend"] == "gloo"
    assert rec["dist_selftest"]["ok"] and rec["dist_selftest"]["ranks"] == 2


def test_rccl_group_of_one_runs_the_collective_wrappers(dev, tmp_path):
    """RCCL itself (backend 'nccl', `device_id=` as bench.py passes it) on the one GPU of the box: a one-rank group runs
    the all-gather / all-reduce wrappers and bench.py's MAX-reduce, object gather and barrier on float64 device
    tensors."""
    codes = run_ranks(shard_worker.rccl_single_rank, 1, (str(tmp_path),), timeout=300)
    assert codes == [0], (codes, _errors(tmp_path))
    assert bool(np.load(tmp_path / "rccl.npz")["ok"])


@pytest.mark.parametrize("reg_wires,read_map", [(3, 1), (3, 0), (4, 0), (4, 1)])
def test_cold_process_runs_the_full_n20_batch_correctly(dev, tmp_path, reg_wires, read_map):
    """Two fresh processes each run BASELINE config 4's batch (n = 20, L = 8, 961 circuits) as their FIRST GPU work:
    every row must sum to 1.  (A stage that read across thread groups without a barrier was right in every warm run
    and wrong in five of eight cold ones.)"""
    from oracle import cpu_port as cp, circuit as oc
    want, digests = None, []
    for trial in range(2):
        codes = run_ranks(shard_worker.cold_batch, 1, (20, 8, read_map, str(tmp_path), reg_wires), timeout=300)
        assert codes == [0], (codes, _errors(tmp_path))
        res = np.load(tmp_path / "cold0.npz")
        assert float(res["worst"]) < 1e-12
        digests.append(str(res["digest"]))
        if want is None and cp.available():          # base row + six shifted rows against the C port (computed once)
            th = res["theta"]
            ts = [th.copy()]
            for p_ in res["picks"]:
                for sgn in (+1, -1):
                    t2 = th.copy(); t2[int(p_)] += sgn * np.pi / 2
                    ts.append(t2)
            want = cp.circuit_probs("hardware_efficient", 20, 8, np.stack(ts))
        if want is not None:
            np.testing.assert_allclose(res["rows"], want, rtol=1e-9, atol=1e-17)
    assert digests[0] == digests[1]                  # the whole 961-row batch: bitwise the same in both cold processes
