"""GPU parity of the drop-in classes (QuantumBornMachine, KSDVariationalInference) against the oracle's
restatement of the reference epoch (ksd_vi_quantum.py:110-161)."""
import io
import os
import contextlib
import math

import numpy as np
import pytest
import torch

from oracle import circuit as oc, stein as os_, ksd as ok
from tensornetworks_amd.bayesian_network import get_sprinkler_network, synthetic_network
from conftest import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda", 0)


def make_vi(bn, lat, obs, n, L, ansatz, device, seed=0, **kw):
    from tensornetworks_amd.ksd_vi_quantum import KSDVariationalInference
    torch.manual_seed(seed)
    return KSDVariationalInference(bn, lat, obs, qbm_num_latent_vars=n, qbm_ansatz_layers=L,
                                   qbm_ansatz_type=ansatz, pytorch_device=device, **kw)


@pytest.mark.parametrize("device", ["cpu", "cuda:0"])
@pytest.mark.parametrize("optimizer_type", ["adam", "sgd"])
def test_five_epoch_trace_sprinkler(dev, device, optimizer_type):
    """Run script defaults (run_sprinkler_quantum_ksd.py:35-43): n=3, L=4, hardware_efficient, lr 0.005,
    clip 10, cosine schedule.  theta lives on `device`; the circuits always run on the GPU."""
    bn = get_sprinkler_network(False)
    lat, obs, x = ['C', 'S', 'R'], ['W'], {'W': 1}
    vi = make_vi(bn, lat, obs, 3, 4, "hardware_efficient", device, seed=11)
    theta0 = vi.born_machine.theta.detach().cpu().numpy().copy()
    assert vi.born_machine.theta.dtype == torch.float32 and theta0.shape == (36,)
    post, _ = bn.get_true_posterior(lat, x)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        hist = vi.train(x, num_epochs=5, lr_born_machine=0.005, verbose=True, true_posterior_for_tvd=post,
                        optimizer_type=optimizer_type)
    out = buf.getvalue()
    assert "Precomputing score functions s_p(x,z)..." in out and "Score functions precomputed." in out
    assert "Epoch 1/5 | KSD:" in out and "| LR:" in out and "| TVD:" in out and "Restoring best parameters" in out
    K = os_.gram_closed_form(golden("sprinkler_w1.npz")["S"], 3)
    tr = ok.EpochTrace("hardware_efficient", 3, 4, K, theta0, lr=0.005, num_epochs=5, clip=10.0,
                       optimizer_type=optimizer_type)
    for _ in range(5):
        tr.step()
    assert set(hist) == {'loss_ksd', 'tvd', 'grad_norm'} and len(hist['loss_ksd']) == 5
    np.testing.assert_allclose(hist['loss_ksd'], tr.history['loss_ksd'], rtol=1e-6)
    np.testing.assert_allclose([float(g) for g in hist['grad_norm']], tr.history['grad_norm'], rtol=1e-5)
    np.testing.assert_allclose(vi.born_machine.theta.detach().cpu().numpy(), tr.history['theta'][-1], rtol=0, atol=2e-6)
    assert all(0 <= t <= 1 for t in hist['tvd'])
    assert vi.born_machine.theta.device.type == torch.device(device).type


@pytest.mark.parametrize("ansatz,n,L,mode", [("hardware_efficient", 5, 2, "dense"), ("all_to_all", 5, 2, "kron"),
                                              ("basic", 6, 2, "dense"), ("hardware_efficient", 8, 4, "kron"),
                                              ("hardware_efficient", 8, 4, "dense"), ("hardware_efficient", 8, 4, "auto")])
def test_step_matches_oracle_synthetic(dev, ansatz, n, L, mode):
    """q, loss AND gradient of one step against the oracle, including BASELINE config 2 (n = 8, L = 4) in the `dense`
    mode that `auto` picks there (reference epoch body ksd_vi_quantum.py:110-161)."""
    bn, lat, obs, x = synthetic_network(n, 0)
    vi = make_vi(bn, lat, obs, n, L, ansatz, "cuda:0", seed=3, gram_mode=mode)
    vi._prepare_stein(x)
    loss, grad, q = vi.ksd_and_grad()
    th = vi.born_machine.theta.detach().double().cpu().numpy()
    q_o = oc.probs(ansatz, n, L, th)
    K_o = os_.gram_closed_form(os_.score_matrix(bn, x, lat, obs), n)
    np.testing.assert_allclose(q.cpu().numpy(), q_o, rtol=1e-10, atol=1e-14)
    assert math.isclose(loss.item(), ok.ksd_loss(K_o, q_o), rel_tol=1e-9)
    if mode == "auto":
        assert vi._K is not None and vi._K.shape == (1 << n, 1 << n)         # auto = dense up to n = 16
    g_o = oc.paramshift_vjp(ansatz, n, L, th, ok.ksd_grad_q(K_o, q_o))
    np.testing.assert_allclose(grad.cpu().numpy(), g_o, rtol=1e-7, atol=1e-9 * np.abs(g_o).max())


def test_autograd_path_equals_fused_path(dev):
    """`loss.backward()` through get_probabilities (the reference's route, ksd_vi_quantum.py:113-150)
    gives the same gradient as the fused device step."""
    bn = get_sprinkler_network(False)
    lat, obs, x = ['C', 'S', 'R'], ['W'], {'W': 1}
    vi = make_vi(bn, lat, obs, 3, 2, "hardware_efficient", "cuda:0", seed=5)
    vi._prepare_stein(x)
    loss_f, grad_f, _ = vi.ksd_and_grad()
    bm = vi.born_machine
    q = bm.get_probabilities().to(torch.float64)
    assert q.requires_grad and q.dtype == torch.float64 and q.shape == (8,)
    loss = torch.sqrt((q @ (vi._K @ q)).clamp(min=1e-12))
    loss.backward()
    assert bm.theta.grad.dtype == torch.float32
    np.testing.assert_allclose(bm.theta.grad.cpu().numpy(), grad_f.cpu().numpy(), rtol=2e-6, atol=1e-6 * grad_f.abs().max().item())
    assert math.isclose(loss.item(), loss_f.item(), rel_tol=1e-10)


def test_born_machine_api(dev):
    from tensornetworks_amd.quantum_born_machine import QuantumBornMachine
    torch.manual_seed(0)
    bm = QuantumBornMachine(4, ansatz_layers=2, ansatz_type="basic", init_method="random").to("cuda:0")
    assert bm.num_ansatz_params == 16 and bm.theta.shape == (16,)
    assert QuantumBornMachine(3, 2, init_method="zero").theta.abs().sum().item() == 0.0
    assert QuantumBornMachine(3, 2, ansatz_type="all_to_all").num_ansatz_params == 18
    d = bm.get_prob_dict()
    assert list(d)[:3] == [(0, 0, 0, 0), (0, 0, 0, 1), (0, 0, 1, 0)] and abs(sum(d.values()) - 1) < 1e-12
    q = np.array(list(d.values()))
    np.testing.assert_allclose(q, oc.probs("basic", 4, 2, bm.theta.detach().double().cpu().numpy()), rtol=1e-10, atol=1e-14)
    s = bm.sample(2000)
    assert s.shape == (2000, 4) and s.dtype == torch.float32 and s.device.type == "cuda"
    assert set(np.unique(s.cpu().numpy()).tolist()) <= {0.0, 1.0}
    idx = (s.cpu().numpy() @ np.array([8, 4, 2, 1])).astype(int)
    emp = np.bincount(idx, minlength=16) / 2000
    assert np.abs(emp - q).max() < 0.06
    z = torch.tensor([[0, 0, 0, 0], [1, 0, 1, 1]], dtype=torch.float32, device="cuda:0")
    lq = bm.get_log_q_z_x(z)
    np.testing.assert_allclose(lq.detach().cpu().numpy(), np.log(np.clip(q[[0, 11]], 1e-9, None)), rtol=1e-9)
    with pytest.raises(ValueError):
        bm.get_log_q_z_x(torch.tensor([[0, 2, 0, 0]], dtype=torch.float32))
    assert bm.sample(0).shape == (0, 4)
    assert bm.get_log_q_z_x(torch.empty(0, 4)).shape == (0,)


def test_trainer_errors_and_lazy_attrs(dev):
    bn = get_sprinkler_network(False)
    vi = make_vi(bn, ['C', 'S', 'R'], ['W'], 3, 1, "hardware_efficient", "cuda:0")
    with pytest.raises(ValueError, match="Keys in x_observation_dict"):
        vi.train({'Q': 1}, 1, 0.01, verbose=False)
    assert vi.num_possible_latent_states == 8 and len(vi.all_latent_states_tuples) == 8
    assert vi.all_latent_states_tuples[5] == (1, 0, 1)
    s = vi._get_precomputed_s_p((1, 0, 1), {'W': 1})
    np.testing.assert_array_equal(s.cpu().numpy(), golden("sprinkler_w1.npz")["S"][5])
    with contextlib.redirect_stdout(io.StringIO()):
        hist = vi.train({'W': 1}, 2, 0.01, verbose=False, use_lr_scheduler=False, optimizer_type="other")
    assert len(hist['tvd']) == 2 and all(np.isnan(hist['tvd']))


def test_full_size_step_n16(dev):
    """One KSD-gradient step at BASELINE config 3 (n=16, L=6), matrix-free and dense agree."""
    n, L = 16, 6
    bn, lat, obs, x = synthetic_network(n, 0)
    vi = make_vi(bn, lat, obs, n, L, "hardware_efficient", "cuda:0", seed=0, gram_mode="kron")
    vi._prepare_stein(x)
    loss_k, grad_k, q = vi.ksd_and_grad()
    assert abs(q.sum().item() - 1) < 1e-12 and grad_k.shape == (288,) and torch.isfinite(grad_k).all()
    vi.gram_mode = "dense"
    vi._prepare_stein(x)
    loss_d, grad_d, _ = vi.ksd_and_grad()
    assert math.isclose(loss_k.item(), loss_d.item(), rel_tol=1e-9)
    np.testing.assert_allclose(grad_k.cpu().numpy(), grad_d.cpu().numpy(), rtol=1e-7, atol=1e-9 * grad_d.abs().max().item())


def _c_port_step(ansatz, n, L, theta, S_host, dense):
    """One KSD-gradient step of the oracle's C port (oracle/cpu_port.c): all 2P + 1 circuits, y = K_p q (dense rows in
    blocks, or the matrix-free Kronecker form), loss = sqrt(max(q.y, 1e-12)), grad_p = (q+_p - q-_p) . y / (2 loss)."""
    from oracle import cpu_port as cp
    N, P = 1 << n, theta.size
    probs, _ = cp.paramshift_probs(ansatz, n, L, theta, 0, P, include_base=True)
    q = np.ascontiguousarray(probs[0])
    if dense:
        y = np.empty(N)
        blk = max(1, min(N, (2 << 30) // (8 * N)))
        for r0 in range(0, N, blk):
            K_rows = cp.gram_rows(S_host, n, 1.0, r0, min(N, r0 + blk))
            y[r0:r0 + blk], _ = cp.gemv_rows(K_rows, n, r0, min(N, r0 + blk), q)
            del K_rows
    else:
        y, _ = cp.kron_matvec(S_host, q, n, 1.0)
    loss = math.sqrt(max(float(q @ y), 1e-12))
    grad = 0.5 * ((probs[1::2] - probs[2::2]) @ (y / loss))
    return q, loss, grad


def _bench_theta(P):
    g = torch.Generator().manual_seed(0)
    return (0.1 * torch.randn(P, generator=g, dtype=torch.float32))       # bench.py's theta0 (`small_random`)


def test_config3_full_step_against_c_port(dev):
    """BASELINE config 3 exactly as bench.py measures it (n = 16, L = 6, dense 2^16 x 2^16 Gram, theta0 of the bench):
    q, the loss and the WHOLE 288-vector gradient of one training step against the oracle's C port -- all 577 circuits
    and the full GEMV on the host (reference: ksd_vi_quantum.py:110-150, quantum_born_machine.py:58-87)."""
    from oracle import cpu_port as cp
    if not cp.available():
        pytest.skip("oracle/_build/libcpu_port.so not built")
    n, L = 16, 6
    bn, lat, obs, x = synthetic_network(n, 0)
    vi = make_vi(bn, lat, obs, n, L, "hardware_efficient", "cuda:0", seed=0, gram_mode="dense")
    P = vi.born_machine.num_ansatz_params
    with torch.no_grad():
        vi.born_machine.theta.copy_(_bench_theta(P).to(vi.born_machine.theta.device))
    vi._prepare_stein(x)
    loss, grad, q = vi.ksd_and_grad()
    th = vi.born_machine.theta.detach().double().cpu().numpy()
    q_o, loss_o, grad_o = _c_port_step("hardware_efficient", n, L, th, vi._S.cpu().numpy(), dense=True)
    np.testing.assert_allclose(q.cpu().numpy(), q_o, rtol=1e-10, atol=1e-17)
    assert abs(loss.item() - loss_o) <= 1e-12 * abs(loss_o), (loss.item(), loss_o)
    np.testing.assert_allclose(grad.cpu().numpy(), grad_o, rtol=1e-8, atol=1e-9 * np.abs(grad_o).max())


def test_config4_full_step_against_c_port(dev):
    """BASELINE config 4 on one GPU (n = 20, L = 8, matrix-free contraction): loss and the whole 480-vector gradient of
    one step against the C port's 961 circuits + Kronecker mat-vec (about half a minute of host time; run once)."""
    from oracle import cpu_port as cp
    if not cp.available():
        pytest.skip("oracle/_build/libcpu_port.so not built")
    n, L = 20, 8
    bn, lat, obs, x = synthetic_network(n, 0)
    vi = make_vi(bn, lat, obs, n, L, "hardware_efficient", "cuda:0", seed=0, gram_mode="kron")
    P = vi.born_machine.num_ansatz_params
    with torch.no_grad():
        vi.born_machine.theta.copy_(_bench_theta(P).to(vi.born_machine.theta.device))
    vi._prepare_stein(x)
    loss, grad, q = vi.ksd_and_grad()
    th = vi.born_machine.theta.detach().double().cpu().numpy()
    S_host = vi._S.cpu().numpy()
    loss_v, grad_v, q_v = loss.item(), grad.cpu().numpy(), q.cpu().numpy()
    del vi
    from tensornetworks_amd import backend
    backend.release_workspaces()
    torch.cuda.empty_cache()
    q_o, loss_o, grad_o = _c_port_step("hardware_efficient", n, L, th, S_host, dense=False)
    np.testing.assert_allclose(q_v, q_o, rtol=1e-9, atol=1e-17)
    assert abs(loss_v - loss_o) <= 1e-11 * abs(loss_o), (loss_v, loss_o)
    np.testing.assert_allclose(grad_v, grad_o, rtol=1e-8, atol=1e-9 * np.abs(grad_o).max())


def test_posterior_table_and_device_tvd(dev):
    """exact posterior and TVD as arrays on the device == the reference's dict forms
    (bayesian_network.py:148-253, utils.py:6-36), and train() tracks the same TVD with either."""
    from tensornetworks_amd.stein_utils import true_posterior_table, tvd_table
    from tensornetworks_amd.utils import calculate_tvd
    bn = get_sprinkler_network(False)
    lat, x = ['C', 'S', 'R'], {'W': 1}
    post_dict, p_obs = bn.get_true_posterior(lat, x)
    g = golden("sprinkler_w1.npz")
    post_t, p_obs_t = true_posterior_table(bn, x, lat, device="cuda:0")
    np.testing.assert_allclose(post_t.cpu().numpy(), np.array(list(post_dict.values())), rtol=1e-14)
    np.testing.assert_allclose(post_t.cpu().numpy(), g["posterior"], rtol=1e-13)
    assert abs(p_obs_t - p_obs) < 1e-15 and abs(post_t.sum().item() - 1) < 1e-14
    bn6, lat6, obs6, x6 = synthetic_network(6, 3)
    pd6, _ = bn6.get_true_posterior(lat6, x6)
    pt6, _ = true_posterior_table(bn6, x6, lat6, device="cuda:0")
    np.testing.assert_allclose(pt6.cpu().numpy(), np.array(list(pd6.values())), rtol=1e-13)
    qv = torch.rand(64, dtype=torch.float64, device="cuda:0"); qv /= qv.sum()
    qd = dict(zip(pd6.keys(), qv.cpu().tolist()))
    assert abs(float(tvd_table(pt6, qv)) - calculate_tvd(pd6, qd)) < 1e-15
    hists = []
    for target in (post_dict, post_t):
        vi = make_vi(bn, lat, ['W'], 3, 2, "hardware_efficient", "cuda:0", seed=3)
        with contextlib.redirect_stdout(io.StringIO()):
            hists.append(vi.train(x, 4, 0.05, verbose=True, true_posterior_for_tvd=target))
    np.testing.assert_allclose(hists[0]['tvd'], hists[1]['tvd'], rtol=0, atol=1e-7)   # dict path prints float32-cast probs
    np.testing.assert_array_equal(hists[0]['loss_ksd'], hists[1]['loss_ksd'])


def test_async_step_equals_sync_step_and_guards_on_device(dev):
    """training_step_async (no host sync; NaN/Inf guard inside the fused optimiser kernel) walks the same
    trajectory as training_step, and a non-finite loss leaves theta and the optimiser state untouched."""
    bn = get_sprinkler_network(False)
    x = {'W': 1}
    runs = []
    for use_async in (False, True):
        vi = make_vi(bn, ['C', 'S', 'R'], ['W'], 3, 2, "hardware_efficient", "cuda:0", seed=11)
        vi._prepare_stein(x, announce=False) if 'announce' in vi._prepare_stein.__code__.co_varnames else vi._prepare_stein(x)
        params, opt, sched = vi.make_optimizer(0.05, 5, True, "adam", (0.9, 0.999))
        losses, norms = [], []
        for _ in range(5):
            if use_async:
                l, g, _ = vi.training_step_async(params, opt, sched, 10.0)
            else:
                l, g, _ = vi.training_step(params, opt, sched, 10.0)
            losses.append(l); norms.append(g)
        runs.append(([float(v) for v in losses], [float(v) for v in norms], vi.born_machine.theta.detach().cpu().numpy().copy(),
                     sched.get_last_lr()[0]))
    assert runs[0][0] == runs[1][0] and runs[0][1] == runs[1][1]
    np.testing.assert_array_equal(runs[0][2], runs[1][2])
    assert runs[0][3] == runs[1][3]
    # device-side guard
    vi = make_vi(bn, ['C', 'S', 'R'], ['W'], 3, 2, "hardware_efficient", "cuda:0", seed=11)
    vi._prepare_stein(x)
    params, opt, sched = vi.make_optimizer(0.05, 5, False, "adam", (0.9, 0.999))
    vi.training_step_async(params, opt, sched, 10.0)
    theta_before = vi.born_machine.theta.detach().clone()
    state_before = {k: v.clone() for k, v in opt.state[params[0]].items() if torch.is_tensor(v)}
    real = vi.ksd_and_grad

    def poisoned():
        loss, grad, q = real()
        return loss * float("inf"), grad, q
    vi.ksd_and_grad = poisoned
    l, _, _ = vi.training_step_async(params, opt, sched, 10.0)
    assert not np.isfinite(float(l))
    assert torch.equal(vi.born_machine.theta.detach(), theta_before)
    for k, v in state_before.items():
        assert torch.equal(opt.state[params[0]][k], v), k


def test_cu_partition_mode_gives_the_same_step(dev):
    """overlap_streams = "partition": circuits and contraction on two CU-masked streams (bornvi_stream_create_cu_range)
    -- same loss, gradient and q as the sequential step, bit for bit (independent computations, only placed differently)."""
    import torch
    from tensornetworks_amd.bayesian_network import synthetic_network
    from tensornetworks_amd.ksd_vi_quantum import KSDVariationalInference
    n, L = 14, 2
    bn, lat, obs, x = synthetic_network(n, seed=1)
    torch.manual_seed(0)
    vi = KSDVariationalInference(bn, lat, obs, qbm_num_latent_vars=n, qbm_ansatz_layers=L, pytorch_device=str(dev),
                                 gram_mode="dense")
    vi._prepare_stein(x)
    vi.overlap_streams = False
    vi.fused_dot = False          # (the overlapped modes store the probabilities and dot them afterwards: compare like with like;
    l0, g0, q0 = vi.ksd_and_grad()   # the fused dot sums in another order -- tests/test_gpu_r3.py holds it to 1e-12)
    vi.overlap_streams = "partition"
    try:
        backend_mod = __import__("tensornetworks_amd.backend", fromlist=["backend"])
        ncu = torch.cuda.get_device_properties(dev).multi_processor_count
        backend_mod.cu_range_stream(dev, 0, ncu // 2)
    except Exception as e:                                    # CU masking refused by this driver / container
        pytest.skip(f"hipExtStreamCreateWithCUMask unavailable: {e}")
    l1, g1, q1 = vi.ksd_and_grad()
    l2, g2, q2 = vi.ksd_and_grad()
    torch.cuda.synchronize()
    for a, b in ((l0, l1), (g0, g1), (q0, q1), (l0, l2), (g0, g2)):
        assert torch.equal(a, b)
    vi.choose_overlap(reps=2)
    assert vi.overlap_choice["chosen"] in ("sequential", "partition")


def test_gram_placement_keeps_an_identical_matrix(dev):
    """Large dense K_p: _prepare_stein builds several copies in fresh memory, keeps the one the contraction streams
    fastest (placement matters on this part) and records the timings; the kept matrix is the same bits."""
    import torch
    from tensornetworks_amd import backend
    from tensornetworks_amd.bayesian_network import synthetic_network
    n = 14                                                       # 2 GiB of K_p
    bn, lat, obs, x = synthetic_network(n, seed=2)
    vi = make_vi(bn, lat, obs, n, 1, "hardware_efficient", str(dev), gram_mode="dense")
    assert vi.gram_placement_tries >= 2  # (on by default: the contraction's rate follows the placement of K_p vs its workspace)
    vi.gram_placement_tries = 3
    vi._prepare_stein(x)
    gp = vi.gram_placement
    pairs = gp["contraction_ms_per_pair"]
    assert gp is not None and 1 <= len(pairs) <= vi.gram_placement_tries ** 2
    assert [p for p in pairs if p[:2] == gp["kept"]][0][2] == min(p[2] for p in pairs)
    ref = backend.stein_gram(vi._S, n, vi.base_kernel_length_scale)
    assert torch.equal(vi._K, ref)
    vi.gram_placement_tries = 1
    vi._stein_key = None
    vi._prepare_stein(x)
    assert vi.gram_placement is None and torch.equal(vi._K, ref)


def test_clip_cast_guard_flags_nonfinite_loss(dev):
    """bornvi_clip_cast_grad_guard: same gradient and norm as bornvi_clip_cast_grad (and as torch's clip_grad_norm_),
    found_inf = 1 exactly for NaN / +-Inf losses -- the device-side form of the reference's "Skipping update" guard."""
    from tensornetworks_amd import backend
    g = torch.randn(288, dtype=torch.float64, device=dev) * 7.0
    g32_ref, norm_ref = backend.clip_cast_grad(g, 10.0)
    t = torch.nn.Parameter(torch.zeros(288, dtype=torch.float32, device=dev))
    t.grad = g.to(torch.float32)
    tn = torch.nn.utils.clip_grad_norm_([t], 10.0)
    for val, want in ((1.25, 0.0), (0.0, 0.0), (-3.0, 0.0), (float("nan"), 1.0), (float("inf"), 1.0), (float("-inf"), 1.0),
                      (1e308, 0.0), (5e-324, 0.0)):
        loss = torch.tensor([val], dtype=torch.float64, device=dev)
        g32, norm, found = backend.clip_cast_grad_guard(g, 10.0, loss)
        assert torch.equal(g32, g32_ref) and torch.equal(norm, norm_ref)
        assert float(found) == want, (val, float(found))
    np.testing.assert_allclose(g32_ref.cpu().numpy(), t.grad.cpu().numpy(), rtol=2e-6)
    np.testing.assert_allclose(float(norm_ref), float(tn), rtol=1e-6)


def test_graphed_step_replays_the_async_step(dev):
    """make_graphed_step: BASELINE config 2's step (n = 8, L = 4, dense) captured once into a HIP graph and replayed --
    the same kernels in the same order, so losses and theta are the bits of the eager training_step_async trajectory
    (Adam + cosine schedule + clip, the NaN/Inf guard on the device)."""
    n, L = 8, 4
    bn, lat, obs, x = synthetic_network(n, 0)
    runs = []
    for graphed in (False, True, "device_adam"):
        vi = make_vi(bn, lat, obs, n, L, "hardware_efficient", "cuda:0", seed=21, gram_mode="dense")
        vi._prepare_stein(x)
        params, opt, sched = vi.make_optimizer(0.01, 9, True, "adam", (0.9, 0.999), capturable=True)
        losses = []
        if graphed:
            step = vi.make_graphed_step(params, opt, sched, 10.0, warmup=3,      # 3 eager steps, then replays
                                        device_adam=(graphed == "device_adam"))
            assert (step.adam is not None) == (graphed == "device_adam")
            for _ in range(6):
                l, gn, q = step()
                losses.append(float(l))
            assert abs(float(q.sum()) - 1.0) < 1e-12 and float(gn) > 0
        else:
            for i in range(9):
                l, gn, q = vi.training_step_async(params, opt, sched, 10.0)
                if i >= 3:
                    losses.append(float(l))
        runs.append((losses, vi.born_machine.theta.detach().cpu().numpy().copy(),
                     float(step.last_lr()) if graphed else float(sched.get_last_lr()[0])))
    # the one-launch clip + Adam + schedule (DeviceAdam): the same update to float32 rounding
    np.testing.assert_allclose(runs[2][0], runs[0][0], rtol=2e-6)
    np.testing.assert_allclose(runs[2][1], runs[0][1], rtol=0, atol=2e-6)
    np.testing.assert_allclose(runs[2][2], runs[0][2], rtol=1e-6)
    assert runs[0][0] == runs[1][0]
    np.testing.assert_array_equal(runs[0][1], runs[1][1])
    assert runs[0][2] == runs[1][2]
    assert runs[0][0][-1] < runs[0][0][0]            # it trains
    with pytest.raises(Exception):
        vi2 = make_vi(bn, lat, obs, n, L, "hardware_efficient", "cuda:0", seed=21, gram_mode="dense")
        vi2._prepare_stein(x)
        p2, o2, s2 = vi2.make_optimizer(0.01, 9, True, "adam", (0.9, 0.999))        # not capturable
        vi2.make_graphed_step(p2, o2, s2, 10.0)


def test_clip_adam_step_is_torch_adam_behind_the_clip_and_the_guard(dev):
    """bornvi_clip_adam_step against what it replaces: clip_cast_grad_guard -> torch's fused Adam (found_inf) ->
    CosineAnnealingLR.step() -> theta.double(), over 12 epochs with a NaN loss and an Inf loss among them (skipped
    updates; the schedule moves on), a clipped and an unclipped gradient, and a schedule table shorter than the run
    (refilled on the way).  Moments, theta and norms to float32 rounding, the epoch / step counts exactly."""
    from tensornetworks_amd import backend
    from tensornetworks_amd.ksd_vi_quantum import DeviceAdam
    P, T, lr0 = 37, 7, 0.05
    g = torch.Generator(device="cpu").manual_seed(5)
    theta0 = torch.randn(P, generator=g)
    grads = [torch.randn(P, generator=g, dtype=torch.float64) * (30.0 if e % 3 == 0 else 0.5) for e in range(12)]
    losses = [float("nan") if e == 4 else float("inf") if e == 9 else 1.0 + e for e in range(12)]
    # torch
    th = torch.nn.Parameter(theta0.clone().to(dev))
    opt = torch.optim.Adam([th], lr=torch.tensor(lr0, device=dev), betas=(0.9, 0.999), fused=True, capturable=True)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=T, eta_min=lr0 / 10)
    ref = []
    for e in range(12):
        g32, norm, found = backend.clip_cast_grad_guard(grads[e].to(dev), 10.0, torch.tensor([losses[e]], dtype=torch.float64, device=dev))
        th.grad = g32
        opt.found_inf = found
        opt.step()
        del opt.found_inf
        sch.step()
        ref.append((th.detach().clone(), float(norm)))
    st = opt.state[th]
    # ours (a table of 5 entries: refilled twice during the run)
    th2 = torch.nn.Parameter(theta0.clone().to(dev))
    adam = DeviceAdam(th2, lr0, (0.9, 0.999), 1e-8, T_max=T, eta_min=lr0 / 10, capacity=5)
    for e in range(12):
        norm = adam.step(grads[e].to(dev), torch.tensor([losses[e]], dtype=torch.float64, device=dev), 10.0)
        adam.advance()
        np.testing.assert_allclose(th2.detach().cpu().numpy(), ref[e][0].cpu().numpy(), rtol=0, atol=3e-7, err_msg=f"epoch {e}")
        assert float(norm) == ref[e][1]
        assert torch.equal(adam.theta64, th2.detach().double())
        if e in (4, 9):
            assert torch.equal(th2.detach(), ref[e - 1][0]) or np.allclose(th2.detach().cpu(), ref[e - 1][0].cpu(), atol=3e-7)
    np.testing.assert_allclose(adam.exp_avg.cpu().numpy(), st["exp_avg"].cpu().numpy(), rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(adam.exp_avg_sq.cpu().numpy(), st["exp_avg_sq"].cpu().numpy(), rtol=2e-6, atol=1e-12)
    assert int(adam.counters[0]) == 10 == int(st["step"]) and adam.epochs == 12
    hl, hn = adam.history()
    np.testing.assert_array_equal(hl.cpu().numpy(), np.array(losses))
    assert hn.cpu().tolist() == [r[1] for r in ref]
    part = adam.history(3, 6)[0].cpu().tolist()
    assert part[0] == 4.0 and np.isnan(part[1]) and part[2] == 6.0 and len(adam.history(3, 6)[1]) == 3
    assert abs(adam.last_lr() - float(sch.get_last_lr()[0])) < 1e-7
    # the clipped gradient is what theta.grad holds
    np.testing.assert_array_equal(th2.grad.cpu().numpy(), g32.cpu().numpy())
    with pytest.raises(backend.BornviError):
        DeviceAdam(torch.zeros(3, dtype=torch.float64, device=dev), 0.1)


def test_graphed_step_writes_only_memory_it_owns(dev):
    """A replay writes through pointers fixed at capture time.  Every tensor behind them has to live as long as step()
    does: one freed after the capture (the guard flag was, once) has its block handed to the caller's next small
    tensor, and the next replay writes 0.0 into that.  Small tensors allocated after the capture keep their values."""
    import gc
    n, L = 8, 2
    bn, lat, obs, x = synthetic_network(n, 5)
    vi = make_vi(bn, lat, obs, n, L, "hardware_efficient", "cuda:0", seed=3, gram_mode="dense")
    vi._prepare_stein(x)
    params, opt, sched = vi.make_optimizer(0.05, 12, True, "adam", (0.9, 0.999), capturable=True)
    step = vi.make_graphed_step(params, opt, sched, 10.0, warmup=2)
    gc.collect()
    canaries = [torch.full((k,), 7.0, dtype=torch.float32, device=dev) for k in (1, 1, 1, 2, 8, 32, 64, 128) for _ in range(16)]
    kept = []
    for _ in range(4):
        kept.append(tuple(t.clone() for t in step()))
        canaries += [torch.full((), 7.0, dtype=torch.float32, device=dev) for _ in range(8)]
    torch.cuda.synchronize()
    assert all(bool((c == 7.0).all()) for c in canaries)
    norms = [float(k[1]) for k in kept]
    assert all(g > 0 for g in norms), norms


@pytest.mark.parametrize("n,L,opt,with_tvd", [(3, 2, "adam", True), (3, 2, "sgd", False), (8, 2, "adam", False), (8, 1, "adam", True)])
def test_train_without_host_sync_returns_the_same_history(dev, n, L, opt, with_tvd, capsys):
    """train(host_sync=False): the reference's epochs without the per-epoch loss.item() -- training_step_async, or for
    n <= 13 with Adam and no per-epoch TVD its HIP-graph replay -- and the values read back at the log points.  Same
    history as train() (the graph path uses torch's capturable Adam: equal to rounding), same final theta, same log lines."""
    from tensornetworks_amd import stein_utils
    if n == 3:
        bn, lat, obs, x = get_sprinkler_network(False), ['C', 'S', 'R'], ['W'], {'W': 1}
    else:
        bn, lat, obs, x = synthetic_network(n, 5)
    runs = []
    for host_sync in (True, False):
        vi = make_vi(bn, lat, obs, n, L, "hardware_efficient", "cuda:0", seed=3, gram_mode="dense")
        post = stein_utils.true_posterior_table(bn, x, lat, dev)[0] if with_tvd else None
        h = vi.train(x, 12, 0.05, verbose=True, true_posterior_for_tvd=post, optimizer_type=opt, host_sync=host_sync)
        out = capsys.readouterr().out
        runs.append((h, vi.born_machine.theta.detach().cpu().numpy().copy(), [l for l in out.splitlines() if l.startswith("Epoch ")]))
    (h0, t0, log0), (h1, t1, log1) = runs
    assert len(h1["loss_ksd"]) == 12 and len(h1["grad_norm"]) == 12 and len(h1["tvd"]) == 12
    np.testing.assert_allclose(h1["loss_ksd"], h0["loss_ksd"], rtol=2e-5)
    np.testing.assert_allclose(h1["grad_norm"], [float(g) for g in h0["grad_norm"]], rtol=2e-4)
    np.testing.assert_allclose(t1, t0, rtol=0, atol=2e-5)
    if with_tvd:
        np.testing.assert_allclose(h1["tvd"], h0["tvd"], rtol=0, atol=1e-5)
    else:
        assert all(np.isnan(v) for v in h1["tvd"])
    assert len(log1) == len(log0) and all(a.split(" | ")[0] == b.split(" | ")[0] for a, b in zip(log0, log1))
    assert h1["loss_ksd"][-1] < h1["loss_ksd"][0]


def test_train_without_host_sync_and_without_a_schedule(dev, capsys):
    """use_lr_scheduler=False through the graph-replayed path: DeviceAdam runs with a constant rate (a one-value table) --
    the same history as the reference-style loop, no LR in the log lines."""
    n, L = 8, 1
    bn, lat, obs, x = synthetic_network(n, 2)
    runs = []
    for host_sync in (True, False):
        vi = make_vi(bn, lat, obs, n, L, "hardware_efficient", "cuda:0", seed=9, gram_mode="dense")
        h = vi.train(x, 8, 0.02, verbose=True, use_lr_scheduler=False, host_sync=host_sync)
        out = capsys.readouterr().out
        runs.append((h, vi.born_machine.theta.detach().cpu().numpy().copy(), [l for l in out.splitlines() if l.startswith("Epoch ")]))
    (h0, t0, log0), (h1, t1, log1) = runs
    np.testing.assert_allclose(h1["loss_ksd"], h0["loss_ksd"], rtol=2e-5)
    np.testing.assert_allclose(h1["grad_norm"], [float(g) for g in h0["grad_norm"]], rtol=2e-4)
    np.testing.assert_allclose(t1, t0, rtol=0, atol=2e-5)
    assert log0 and len(log1) == len(log0) and not any("LR:" in l for l in log0 + log1)


def test_second_train_keeps_the_gram_matrix_unless_the_scores_change(dev, capsys):
    """K_p depends on (S, n, length scale) only: a second _prepare_stein / train() on the same observation keeps the matrix
    (at n = 16 that is 32 GiB and a placement search); another observation, length scale or contraction layout rebuilds."""
    n = 9
    bn, lat, obs, x = synthetic_network(n, 4)
    vi = make_vi(bn, lat, obs, n, 1, "basic", "cuda:0", gram_mode="dense")
    vi._prepare_stein(x)
    K0, ptr0 = vi._K, vi._K.data_ptr()
    vi.train(x, 2, 0.01, verbose=False)
    assert vi._K.data_ptr() == ptr0                       # same observation: kept
    x2 = {k: 1 - v for k, v in x.items()}
    vi._prepare_stein(x2)
    assert not torch.equal(vi._K, K0)                     # other observation: other scores, rebuilt
    vi._prepare_stein(x)
    assert torch.equal(vi._K, K0)
    ptr1 = vi._K.data_ptr()
    vi.base_kernel_length_scale = 2.0
    vi._prepare_stein(x)
    assert not torch.equal(vi._K, K0)
    vi.base_kernel_length_scale = 1.0
    vi.symmetric_contraction = False
    vi._prepare_stein(x)
    assert torch.equal(vi._K, K0)
    capsys.readouterr()


def test_example_driver_runs(dev):
    """examples/run_sprinkler_quantum_ksd.py -- the reference's run script on this engine -- end to end in a fresh process."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "examples", "run_sprinkler_quantum_ksd.py"), "--epochs", "30", "--quiet"],
                         capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "Final TVD:" in out.stdout and "(1, 0, 1)" in out.stdout and "30 epochs in" in out.stdout
