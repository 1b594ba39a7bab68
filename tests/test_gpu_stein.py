"""GPU parity of the Stein side (score, Gram, quadratic form, matrix-free mat-vec, per-element API)
against the golden vectors captured from the reference and against the CPU oracle."""
import math
from functools import partial

import numpy as np
import pytest
import torch

from oracle import stein as os_, ksd as ok
from tensornetworks_amd.bayesian_network import (BayesianNetwork, get_sprinkler_network, pack_network,
                                                  synthetic_network)
from conftest import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def be():
    from tensornetworks_amd import backend
    return backend


def two_node():
    bn = BayesianNetwork()
    bn.add_node('A', cpt={(): {0: 0.8, 1: 0.2}})
    bn.add_node('B', cpt={(0,): {0: 0.7, 1: 0.3}, (1,): {0: 0.4, 1: 0.6}}, parent_names=['A'])
    return bn


@pytest.mark.parametrize("name", ["sprinkler_w1", "sprinkler_w0", "sprinkler_rand0", "sprinkler_rand1",
                                  "sprinkler_rand2", "synthetic_n4_s0", "synthetic_n5_s0", "synthetic_n5_s1",
                                  "synthetic_n6_s0", "synthetic_n8_s0", "two_node"])
def test_score_and_gram_golden(be, dev, name):
    """S bit-exact (same fp64 operation order as the reference); K_p within 3e-15 of max|K|."""
    g = golden(name + ".npz")
    n = g["S"].shape[1]
    if name == "two_node":
        packed = pack_network(two_node(), ['A'], {'B': 1})
    else:
        packed = {k[5:]: g[k] for k in g.files if k.startswith("pack_")}
    S, pxz = be.score_from_packed(packed, n, dev)
    np.testing.assert_array_equal(S.cpu().numpy(), g["S"])
    np.testing.assert_array_equal(pxz.cpu().numpy(), g["pxz"])
    K = be.stein_gram(S, n, 1.0).cpu().numpy()
    rows = g["rows"] if "rows" in g.files else np.arange(2 ** n)
    np.testing.assert_allclose(K[rows], g["K"], rtol=0, atol=3e-15 * np.abs(g["K"]).max())
    assert np.array_equal(K, K.T)                       # evaluated symmetrically: bitwise symmetric


def test_hidden_nodes_are_marginalised(be, dev):
    """Latents (C, R), observed W, Sprinkler node S summed out (stein_utils.py:95-111)."""
    bn = get_sprinkler_network(False)
    lat, x = ['C', 'R'], {'W': 1}
    S, pxz = be.score_from_packed(pack_network(bn, lat, x), 2, dev)
    np.testing.assert_array_equal(S.cpu().numpy(), os_.score_matrix(bn, x, lat))
    np.testing.assert_array_equal(pxz.cpu().numpy(), os_.joint_vector(bn, x, lat))


def test_degenerate_score_is_zero(be, dev):
    """|p(x,z)| < 1e-12 -> zero score vector (stein_utils.py:126-128)."""
    bn = BayesianNetwork()
    bn.add_node('A', cpt={(): {0: 1.0, 1: 0.0}})
    bn.add_node('B', cpt={(0,): {0: 0.5, 1: 0.5}, (1,): {0: 0.5, 1: 0.5}}, parent_names=['A'])
    S, pxz = be.score_from_packed(pack_network(bn, ['A'], {'B': 1}), 1, dev)
    So = os_.score_matrix(bn, {'B': 1}, ['A'])
    np.testing.assert_array_equal(S.cpu().numpy(), So)
    assert S[1, 0].item() == 0.0


@pytest.mark.parametrize("n,seed", [(3, 0), (6, 0), (9, 1), (11, 2)])
def test_quadform_and_kron_match_oracle(be, dev, n, seed):
    bn, lat, obs, x = synthetic_network(n, seed)
    S, pxz = be.score_from_packed(pack_network(bn, lat, x), n, dev)
    So = S.cpu().numpy()
    Ko = os_.gram_closed_form(So, n) if n <= 9 else None
    K = be.stein_gram(S, n, 1.0)
    if Ko is not None:
        np.testing.assert_allclose(K.cpu().numpy(), Ko, rtol=0, atol=3e-15 * np.abs(Ko).max())
    Kn = K.cpu().numpy()
    rng = np.random.default_rng(seed)
    Q = rng.random((3, 2 ** n)); Q /= Q.sum(1, keepdims=True)
    Qt = torch.as_tensor(Q, device=dev)
    ksd2, Y = be.stein_quadform(K, Qt, n)
    scale = np.abs(Kn).max()
    np.testing.assert_allclose(Y.cpu().numpy(), Q @ Kn.T, rtol=0, atol=1e-13 * scale)
    for b in range(3):
        ref = Q[b] @ Kn @ Q[b]
        tol = 1e-13 * float(np.abs(Q[b][:, None] * Q[b][None, :] * Kn).sum())
        assert abs(ksd2[b].item() - ref) < tol
        k2, y = be.stein_matvec_kron(S, Qt[b].contiguous(), n, 1.0)
        np.testing.assert_allclose(y.cpu().numpy(), Kn @ Q[b], rtol=0, atol=1e-12 * scale)
        assert abs(k2.item() - ref) < 10 * tol
    # only ksd2 requested
    k_only, none = be.stein_quadform(K, Qt, n, want_y=False)
    assert none is None
    np.testing.assert_array_equal(k_only.cpu().numpy(), ksd2.cpu().numpy())


@pytest.mark.parametrize("n", [1, 2, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14])
def test_symmetric_quadform_equals_full(be, dev, n):
    """Upper-triangle contraction == full-matrix contraction (K_p is bitwise symmetric)."""
    bn, lat, obs, x = synthetic_network(n, 1)
    S, _ = be.score_from_packed(pack_network(bn, lat, x), n, dev)
    K = be.stein_gram(S, n, 1.0)
    g = torch.Generator().manual_seed(n)
    q = torch.rand(2 ** n, generator=g, dtype=torch.float64).to(dev)
    q /= q.sum()
    k_full, Y = be.stein_quadform(K, q, n)
    k_sym, y_sym = be.stein_quadform_sym(K, q, n)
    scale = (K.abs() @ q).max().item()
    assert (Y[0] - y_sym).abs().max().item() <= 1e-13 * scale
    assert abs(k_full.item() - k_sym.item()) <= 1e-13 * float((q[:, None] * q[None, :] * K).abs().sum())
    # the reference value: dense NumPy product
    np.testing.assert_allclose(y_sym.cpu().numpy(), K.cpu().numpy() @ q.cpu().numpy(), rtol=0, atol=1e-13 * scale)
    if n >= 9:
        # a padded row pitch (the [:, :2^n] view of a [2^n, 2^n + 32] buffer) holds the same matrix.  Below n = 14 a dense
        # matrix takes the full-matrix kernel and a padded one the band kernel (two summation orders: equal to rounding);
        # each is deterministic from call to call
        Kp = be.stein_gram(S, n, 1.0, ld=2 ** n + 32)
        assert Kp.stride(0) == 2 ** n + 32 and torch.equal(Kp, K)
        k_pad, y_pad = be.stein_quadform_sym(Kp, q, n)
        assert (y_pad - y_sym).abs().max().item() <= 1e-13 * scale
        assert abs(k_pad.item() - k_sym.item()) <= 1e-13 * float((q[:, None] * q[None, :] * K).abs().sum())
        np.testing.assert_allclose(y_pad.cpu().numpy(), K.cpu().numpy() @ q.cpu().numpy(), rtol=0, atol=1e-13 * scale)
        k_pad2, y_pad2 = be.stein_quadform_sym(Kp, q, n)
        assert torch.equal(y_pad2, y_pad) and torch.equal(k_pad2, k_pad)
        k_again, y_again = be.stein_quadform_sym(K, q, n)
        assert torch.equal(y_again, y_sym) and torch.equal(k_again, k_sym)


@pytest.mark.parametrize("n,world", [(9, 1), (9, 2), (10, 2), (10, 3), (12, 8), (13, 4), (14, 8)])
def test_sym_strip_pair_shard_sums_to_full(be, dev, n, world):
    """Several GPUs: each rank holds two row blocks of K_p (its strip pairs of the upper triangle) and computes
    an additive share of (K q, q^T K q); the shares (what the all-reduce sums) add up to the full contraction."""
    bn, lat, obs, x = synthetic_network(n, 2)
    S, _ = be.score_from_packed(pack_network(bn, lat, x), n, dev)
    K = be.stein_gram(S, n, 1.0)
    g = torch.Generator().manual_seed(n)
    q = torch.rand(2 ** n, generator=g, dtype=torch.float64).to(dev)
    q /= q.sum()
    k_ref, y_ref = be.stein_quadform_sym(K, q, n)
    total = torch.zeros(2 ** n + 1, dtype=torch.float64, device=dev)
    rows_seen = 0
    for rank in range(world):
        sp = be.sym_pair_shard(n, rank, world)
        assert sp is not None
        (pa, pb), (l0, l1), (h0, h1) = sp
        K_lo = be.stein_gram(S, n, 1.0, rows=(l0, l1)) if l1 > l0 else None
        K_hi = be.stein_gram(S, n, 1.0, rows=(h0, h1)) if h1 > h0 else None
        if l1 > l0:       # the blocks a rank builds are the rows of the full matrix, bit for bit
            assert torch.equal(K_lo, K[l0:l1]) and torch.equal(K_hi, K[h0:h1])
        rows_seen += (l1 - l0) + (h1 - h0)
        total += be.stein_quadform_sym_pairs(K_lo, K_hi, pa, pb, q, n)
    assert rows_seen == 2 ** n
    scale = (K.abs() @ q).max().item()
    assert (total[:-1] - y_ref).abs().max().item() <= 1e-13 * scale
    assert abs(total[-1].item() - k_ref.item()) <= 1e-13 * float((q[:, None] * q[None, :] * K).abs().sum())
    assert be.sym_pair_shard(8, 0, 2) is None       # 256 outcomes: one band of 256 rows, no pairs to deal out


def test_length_scale(be, dev):
    g = golden("synthetic_n5_s0.npz")
    S = torch.as_tensor(g["S"], device=dev)
    for ls in (0.5, 2.0):
        K = be.stein_gram(S, 5, ls).cpu().numpy()
        np.testing.assert_allclose(K, os_.gram_closed_form(g["S"], 5, ls), rtol=0, atol=3e-15 * np.abs(K).max())
        np.testing.assert_allclose(K, os_.gram_loop(g["S"], 5, ls), rtol=0, atol=1e-12 * np.abs(K).max())
    from tensornetworks_amd._ext import BornviError
    with pytest.raises(BornviError):
        be.stein_gram(S, 5, 0.0)


def test_reference_known_answers_through_dropin_api(dev):
    """The seven assertions of stein_utils.py:205-251 against the drop-in functions."""
    from tensornetworks_amd import stein_utils as su
    t64 = torch.float64
    assert su.flip_bit((0, 0, 0), 0) == (1, 0, 0)
    z1 = torch.tensor([0, 0, 1, 1], dtype=t64); z2 = torch.tensor([1, 0, 0, 1], dtype=t64)
    assert su.hamming_distance_torch(z1, z2).item() == 2.0
    assert torch.isclose(su.base_hamming_kernel_torch(z1, z2, 4, length_scale=1.0), torch.tensor(np.exp(-0.5), dtype=t64))
    bn = two_node()
    lat, obs, x = ['A'], ['B'], {'B': 1}
    assert np.isclose(su.compute_prob_joint_xz(bn, x, (1,), lat, obs), 0.12)
    s1 = su.get_score_function_sp_for_z(bn, x, (1,), lat, obs, device='cpu')
    s0 = su.get_score_function_sp_for_z(bn, x, (0,), lat, obs, device='cpu')
    assert s1.dtype == t64 and torch.isclose(s1[0], torch.tensor(-1.0, dtype=t64))
    assert torch.isclose(s0[0], torch.tensor(0.5, dtype=t64))
    kf = partial(su.base_hamming_kernel_torch, num_vars=1, length_scale=1.0)
    kp01 = su.get_stein_kernel_kp_value((0,), (1,), x, bn, lat, obs, kf, s0, s1, device='cpu')
    kp00 = su.get_stein_kernel_kp_value((0,), (0,), x, bn, lat, obs, kf, s0, s0, device='cpu')
    assert torch.isclose(kp01, torch.tensor(2 * np.exp(-1.0) - 2.5, dtype=t64))
    assert torch.isclose(kp00, torch.tensor(1.25 - np.exp(-1.0), dtype=t64))
    with pytest.raises(TypeError):
        su.get_stein_kernel_kp_value((0,), (1,), x, bn, lat, obs, lambda a, b: 1.0, s0, s1)
    # special cases of the base kernel (:36-40, :52-54)
    assert su.base_hamming_kernel_torch(torch.zeros(0), torch.zeros(0), 0).item() == 1.0
    assert su.base_hamming_kernel_torch(z1, z1, 4, length_scale=0.0).item() == 1.0
    assert su.base_hamming_kernel_torch(z1, z2, 4, length_scale=0.0).item() == 0.0


def test_kp_pairs_match_gram(be, dev):
    g = golden("synthetic_n6_s0.npz")
    S = torch.as_tensor(g["S"], device=dev)
    rng = np.random.default_rng(0)
    zi = torch.as_tensor(rng.integers(0, 64, 200), device=dev)
    zj = torch.as_tensor(rng.integers(0, 64, 200), device=dev)
    out = be.stein_kp_pairs(6, 1.0, zi, zj, S[zi].contiguous(), S[zj].contiguous()).cpu().numpy()
    np.testing.assert_allclose(out, g["K"][zi.cpu().numpy(), zj.cpu().numpy()], rtol=0, atol=3e-15 * np.abs(g["K"]).max())


def test_full_size_gram_properties(be, dev):
    """n = 16 (BASELINE config 3): the 32 GiB dense Gram against size-independent properties:
    K p(z|x) ~ 0, q^T K q from the dense GEMV == the matrix-free Kronecker mat-vec, sampled
    entries == the per-pair kernel, sampled rows symmetric."""
    n = 16
    bn, lat, obs, x = synthetic_network(n, 0)
    S, pxz = be.score_from_packed(pack_network(bn, lat, x), n, dev)
    K = be.stein_gram(S, n, 1.0)
    post = (pxz / pxz.sum()).contiguous()
    k2p, yp = be.stein_quadform(K, post, n)
    g = torch.Generator().manual_seed(5)
    rows = torch.randint(0, 2 ** n, (2048,), generator=g).to(dev)
    cancel_scale = (K[rows].abs() @ post).cpu().numpy()  # sum_j |K_ij| p_j: what each y_i is a difference of
    # (K p is not ~0 here: states with p(x,z) < 1e-12 get a zero score row, stein_utils.py:126-128)
    y_oracle = os_.stein_matvec_kron(S.cpu().numpy(), post.cpu().numpy(), n)       # CPU, matrix-free
    r = rows.cpu().numpy()
    assert (np.abs(yp[0].cpu().numpy()[r] - y_oracle[r]) <= 1e-12 * cancel_scale).all()
    assert abs(k2p.item()) <= yp.abs().max().item()     # sum_i p_i y_i with sum p = 1
    q = torch.rand(2 ** n, generator=g, dtype=torch.float64).to(dev)
    q /= q.sum()
    k2d, yd = be.stein_quadform(K, q, n)
    k2s, ys = be.stein_quadform_sym(K, q, n)
    assert abs(k2d.item() - k2s.item()) <= 1e-12 * abs(k2d.item())
    assert (yd[0] - ys).abs().max().item() <= 1e-12 * yd.abs().max().item()
    k2k, yk = be.stein_matvec_kron(S, q, n, 1.0)
    assert abs(k2d.item() - k2k.item()) <= 1e-9 * abs(k2d.item())
    assert (yd[0] - yk).abs().max().item() <= 1e-10 * yd.abs().max().item()
    idx = torch.randint(0, 2 ** n, (4096,), generator=g).to(dev)
    jdx = torch.randint(0, 2 ** n, (4096,), generator=g).to(dev)
    pairs = be.stein_kp_pairs(n, 1.0, idx, jdx, S[idx].contiguous(), S[jdx].contiguous())
    assert (pairs - K[idx, jdx]).abs().max().item() <= 1e-13 * max(pairs.abs().max().item(), 1.0)
    assert torch.equal(K[idx, jdx], K[jdx, idx])


def test_wrappers_refuse_wrong_sizes(dev):
    """The C ABI takes raw pointers: the tensor wrappers check every element count the kernels will index and raise
    BornviError instead of launching (an out-of-bounds device read otherwise)."""
    from tensornetworks_amd import backend
    from tensornetworks_amd._ext import BornviError
    n = 6
    N = 1 << n
    S = torch.zeros((N, n), dtype=torch.float64, device=dev)
    q = torch.full((N,), 1.0 / N, dtype=torch.float64, device=dev)
    K = backend.stein_gram(S, n, 1.0)
    bad_calls = [
        lambda: backend.stein_gram(S[: N // 2].contiguous(), n, 1.0),                      # S of a smaller network
        lambda: backend.stein_gram(S, n, 1.0, rows=(0, N + 1)),
        lambda: backend.stein_quadform_sym(K[: N // 2].contiguous(), q, n),
        lambda: backend.stein_quadform_sym(K, q[:-2].contiguous(), n),
        lambda: backend.stein_quadform(K, q[:-1].contiguous(), n),
        lambda: backend.stein_quadform_rows(K[:8].contiguous(), 0, 16, q, n),
        lambda: backend.stein_quadform_rows(K[:8].contiguous(), 0, 8, q[:-1].contiguous(), n),
        lambda: backend.stein_matvec_kron(S, q[:-1].contiguous(), n, 1.0),
        lambda: backend.stein_matvec_kron(S[:, :-1].contiguous(), q, n, 1.0),
        lambda: backend.ksd_grad_finish(n, torch.zeros((3, N), dtype=torch.float64, device=dev), 2, q, q[:1].contiguous()),
        lambda: backend.ksd_grad_finish(n, torch.zeros((4, N), dtype=torch.float64, device=dev), 2, q[:-1].contiguous(), q[:1].contiguous()),
        lambda: backend.paramshift_probs("basic", n, 2, torch.zeros(5, dtype=torch.float64, device=dev), 0, 1),
        lambda: backend.paramshift_grad("basic", n, 2, torch.zeros(24, dtype=torch.float64, device=dev), q[:-1].contiguous(), 0, 1),
    ]
    for i, call in enumerate(bad_calls):
        with pytest.raises(BornviError):
            call()
            pytest.fail(f"call {i} was executed")
    ksd2, y = backend.stein_quadform_sym(K, q, n)             # the well-formed calls still run
    assert y.shape == (N,) and ksd2.shape == (1,)


@pytest.mark.parametrize("n,B", [(8, 2), (8, 129), (10, 5), (11, 577), (12, 130)])
def test_batched_quadform_on_matrix_cores(be, dev, n, B):
    """bornvi_stein_quadform with B > 1: one pass over K_p with v_mfma_f64_16x16x4 (kernels_batched.hip) against the
    looped HBM-bound GEMV (option batched_quadform = 0) and a dense fp64 product: Y to 1e-13 * sum |terms|, ksd2 likewise;
    vector counts that are not multiples of the 128-vector block; Y optional (reference accumulation
    ksd_vi_quantum.py:123-145, evaluated for B distributions at once)."""
    bn, lat, obs, x = synthetic_network(n, 4)
    S, _ = be.score_from_packed(pack_network(bn, lat, x), n, dev)
    K = be.stein_gram(S, n, 1.0)
    g = torch.Generator().manual_seed(100 * n + B)
    Q = torch.rand((B, 2 ** n), generator=g, dtype=torch.float64).to(dev)
    Q /= Q.sum(dim=1, keepdim=True)
    try:
        be.set_engine_option(dev, "batched_quadform", 0)
        k_loop, Y_loop = be.stein_quadform(K, Q, n)
        be.set_engine_option(dev, "batched_quadform", 1)
        k_mfma, Y_mfma = be.stein_quadform(K, Q, n)
        k_only, none = be.stein_quadform(K, Q, n, want_y=False)
    finally:
        be.set_engine_option(dev, "batched_quadform", 1)
    terms = (K.abs() @ Q.t()).t()                       # sum_j |K_ij| q_bj
    assert ((Y_mfma - Y_loop).abs() <= 1e-13 * terms).all()
    assert ((Y_mfma - (K @ Q.t()).t()).abs() <= 1e-13 * terms).all()
    scale = (Q * terms).sum(dim=1)
    assert ((k_mfma - k_loop).abs() <= 1e-13 * scale).all()
    assert none is None and torch.equal(k_only, k_mfma)
    k_again, Y_again = be.stein_quadform(K, Q, n)       # deterministic
    assert torch.equal(Y_again, Y_mfma) and torch.equal(k_again, k_mfma)


def test_batched_quadform_on_the_trainers_padded_gram(be, dev):
    """bornvi_stein_quadform_ld: the batched matrix-core contraction reads a K_p with a row pitch (the trainer builds
    K_p with pitch 2^n + 32 for n >= 14, bornvi_stein_gram_ld) -- the KSD at every shifted point needs no second dense
    copy.  Y and ksd2 bitwise equal to the run on the dense copy of the same matrix; a padded K without the batched form
    (B = 1, or batched_quadform = 0) is refused, not misread."""
    from tensornetworks_amd._ext import BornviError
    n, B = 14, 130
    N = 1 << n
    bn, lat, obs, x = synthetic_network(n, 4)
    S, _ = be.score_from_packed(pack_network(bn, lat, x), n, dev)
    ld = be.gram_ld(n)
    assert ld == N + 32
    Kp = be.stein_gram(S, n, 1.0, ld=ld)
    assert Kp.shape == (N, N) and Kp.stride(0) == ld
    Kd = Kp.contiguous()
    g = torch.Generator().manual_seed(5)
    Q = torch.rand((B, N), generator=g, dtype=torch.float64).to(dev)
    Q /= Q.sum(dim=1, keepdim=True)
    k_d, Y_d = be.stein_quadform(Kd, Q, n)
    k_p, Y_p = be.stein_quadform(Kp, Q, n)
    assert torch.equal(Y_p, Y_d) and torch.equal(k_p, k_d)
    with pytest.raises(BornviError, match="padded K"):
        be.stein_quadform(Kp, Q[0], n)
    try:
        be.set_engine_option(dev, "batched_quadform", 0)
        with pytest.raises(BornviError, match="padded K"):
            be.stein_quadform(Kp, Q, n)
    finally:
        be.set_engine_option(dev, "batched_quadform", 1)
    # the symmetric contraction refuses a pitch where it runs the dense full-matrix kernel (n < 9)
    K8 = torch.zeros((256, 256 + 32), dtype=torch.float64, device=dev)[:, :256]
    with pytest.raises(BornviError, match="padded K"):
        be.stein_quadform_sym(K8, torch.full((256,), 1 / 256, dtype=torch.float64, device=dev), 8)
