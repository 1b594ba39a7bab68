"""Size-independent properties of the hot path on the GPU, on randomly drawn cases (hypothesis, derandomised):
SURVEY.md section 4 (iv) -- sum q = 1, K_p symmetric PSD, K_p p_true = 0 (so KSD(p_true) = 0), parameter shift ==
finite difference, linearity of the contraction -- each through the C ABI, each also compared with the CPU oracle."""
import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

from oracle import circuit as oc, stein as os_
from tensornetworks_amd.bayesian_network import pack_network, synthetic_network

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
SETTINGS = dict(max_examples=12, deadline=None, derandomize=True)


def _be():
    from tensornetworks_amd import backend
    return backend


@settings(**SETTINGS)
@given(st.sampled_from(oc.ANSATZ_TYPES), st.integers(1, 14), st.integers(0, 4), st.integers(0, 10 ** 6))
def test_probabilities_are_a_distribution_and_match_the_oracle(ansatz, n, L, seed):
    be = _be()
    P = oc.num_params(ansatz, n, L)
    th = np.random.default_rng(seed).uniform(-2 * np.pi, 2 * np.pi, (2, P))
    q = be.circuit_probs(ansatz, n, L, torch.as_tensor(th, device=DEV).contiguous()).cpu().numpy()
    assert q.min() >= 0.0 and np.abs(q.sum(axis=1) - 1.0).max() < 1e-12
    if n <= 11:
        for b in range(2):
            np.testing.assert_allclose(q[b], oc.probs(ansatz, n, L, th[b]), rtol=1e-10, atol=1e-14)


@settings(**SETTINGS)
@given(st.sampled_from(oc.ANSATZ_TYPES), st.integers(1, 9), st.integers(1, 3), st.integers(0, 10 ** 6))
def test_parameter_shift_equals_finite_difference(ansatz, n, L, seed):
    """d/dtheta_p sum_z w_z q_z(theta): the 2P shifted circuits against a central difference of the same engine
    (h = 1e-5: truncation ~ h^2 |q'''| ~ 1e-10) and against the oracle's parameter-shift gradient."""
    be = _be()
    rng = np.random.default_rng(seed)
    P = oc.num_params(ansatz, n, L)
    th = rng.uniform(-1.5, 1.5, P)
    w = rng.normal(size=2 ** n)
    tht, wt = torch.as_tensor(th, device=DEV), torch.as_tensor(w, device=DEV)
    g = be.paramshift_grad(ansatz, n, L, tht, wt, 0, P).cpu().numpy()
    np.testing.assert_allclose(g, oc.paramshift_vjp(ansatz, n, L, th, w), rtol=1e-9, atol=1e-12 * max(1.0, np.abs(g).max()))
    h = 1e-5
    for p in sorted(set(rng.integers(0, P, size=min(P, 4)).tolist())):
        tp, tm = th.copy(), th.copy()
        tp[p] += h; tm[p] -= h
        qq = be.circuit_probs(ansatz, n, L, torch.as_tensor(np.stack([tp, tm]), device=DEV)).cpu().numpy()
        fd = float(w @ (qq[0] - qq[1])) / (2 * h)
        assert abs(fd - g[p]) <= 2e-6 * max(1.0, np.abs(w).sum() * 1e-3 + abs(g[p]))


@settings(**SETTINGS)
@given(st.integers(1, 9), st.integers(0, 10 ** 4), st.floats(0.3, 3.0))
def test_gram_is_symmetric_psd_and_annihilates_the_posterior(n, seed, length_scale):
    be = _be()
    bn, lat, obs, x = synthetic_network(n, seed=seed)
    S, pxz = be.score_from_packed(pack_network(bn, lat, x), n, DEV)
    K = be.stein_gram(S, n, length_scale)
    Kh = K.cpu().numpy()
    assert np.array_equal(Kh, Kh.T)
    np.testing.assert_allclose(Kh, os_.gram_closed_form(S.cpu().numpy(), n, length_scale), rtol=0, atol=1e-13 * np.abs(Kh).max())
    ev = np.linalg.eigvalsh(Kh)
    assert ev.min() > -1e-9 * ev.max()
    p = pxz / pxz.sum()                                     # the exact posterior p(z | x)
    ksd2, y = be.stein_quadform_sym(K, p, n)
    scale = float((K.abs() @ p).max())
    assert float(y.abs().max()) <= 1e-9 * scale and abs(float(ksd2)) <= 1e-9 * scale
    if n >= 2:                                              # matrix-free form: same K p
        k2k, yk = be.stein_matvec_kron(S, p, n, length_scale)
        assert float(yk.abs().max()) <= 1e-9 * scale and abs(float(k2k)) <= 1e-9 * scale


@settings(**SETTINGS)
@given(st.integers(8, 11), st.integers(0, 10 ** 4), st.data())
def test_gram_row_ranges_are_the_rows_of_the_full_matrix(n, seed, data):
    """Any row range [r0, r1) of the matrix-core Gram builder, aligned to its 64-row blocks or not, with or without a padded
    pitch: bit for bit the rows of the full matrix (the row shard of the contraction, and K_p's bitwise symmetry across
    the ranks' blocks, rely on it), and nothing outside the range is written."""
    be = _be()
    N = 1 << n
    g = torch.Generator(device="cpu").manual_seed(seed)
    S = (torch.rand((N, n), generator=g, dtype=torch.float64) * 2 - 1).to(DEV)
    K = be.stein_gram(S, n, 1.0)
    assert torch.equal(K, K.T)
    r0 = data.draw(st.integers(0, N - 1))
    r1 = data.draw(st.integers(r0, N))
    pad = data.draw(st.sampled_from([0, 2, 32]))
    buf = torch.full((r1 - r0 + 2, N + pad), -7.0, dtype=torch.float64, device=DEV)
    out = buf[1:1 + (r1 - r0), :N]
    be.stein_gram(S, n, 1.0, rows=(r0, r1), out=out)
    assert torch.equal(out, K[r0:r1])
    assert bool((buf[0] == -7.0).all()) and bool((buf[-1] == -7.0).all()) and (pad == 0 or bool((buf[:, N:] == -7.0).all()))


@settings(**SETTINGS)
@given(st.integers(2, 13), st.integers(0, 10 ** 4))
def test_contraction_is_linear_and_its_forms_agree(n, seed):
    """K (a u + b v) = a K u + b K v for the symmetric, full-matrix and matrix-free contractions; q^T K q >= 0."""
    be = _be()
    bn, lat, obs, x = synthetic_network(n, seed=seed)
    S, _ = be.score_from_packed(pack_network(bn, lat, x), n, DEV)
    K = be.stein_gram(S, n, 1.0)
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(2 ** n, generator=g, dtype=torch.float64).to(DEV)
    v = torch.rand(2 ** n, generator=g, dtype=torch.float64).to(DEV)
    a, b = 0.37, -1.9
    scale = float((K.abs() @ (u + v)).max())
    forms = {"sym": lambda q: be.stein_quadform_sym(K, q, n),
             "full": lambda q: tuple(t if i == 0 else t[0] for i, t in enumerate(be.stein_quadform(K, q, n, want_y=True))),
             "kron": lambda q: be.stein_matvec_kron(S, q, n, 1.0)}
    ref = None
    for name, f in forms.items():
        k_u, y_u = f(u)
        _, y_v = f(v)
        _, y_c = f(a * u + b * v)
        assert float((y_c - (a * y_u + b * y_v)).abs().max()) <= 1e-12 * scale, name
        assert float(k_u) >= -1e-12 * scale, name
        if ref is None:
            ref = y_u
        else:
            assert float((y_u - ref).abs().max()) <= 1e-11 * scale, name


@pytest.mark.parametrize("ansatz", oc.ANSATZ_TYPES)
@pytest.mark.parametrize("n", [1, 3, 14])
def test_zero_layer_circuits_through_every_entry_point(ansatz, n):
    """No parameters at all (layers = 0): probabilities, the parameter-shift batch (base row only), the gradient
    (empty) and the drop-in class all work -- the Hadamard layer of two ansaetze, no gate for `basic`."""
    be = _be()
    from tensornetworks_amd.quantum_born_machine import QuantumBornMachine
    expect = oc.probs(ansatz, n, 0, np.zeros(0))
    th = torch.zeros(0, dtype=torch.float64, device=DEV)
    q = be.circuit_probs(ansatz, n, 0, torch.zeros((2, 0), dtype=torch.float64, device=DEV)).cpu().numpy()
    np.testing.assert_allclose(q, np.tile(expect, (2, 1)), rtol=0, atol=1e-15)
    rows = be.paramshift_probs(ansatz, n, 0, th, 0, 0, include_base=True).cpu().numpy()
    assert rows.shape == (1, 2 ** n)
    np.testing.assert_allclose(rows[0], expect, rtol=0, atol=1e-15)
    g = be.paramshift_grad(ansatz, n, 0, th, torch.ones(2 ** n, dtype=torch.float64, device=DEV), 0, 0)
    assert g.numel() == 0
    bm = QuantumBornMachine(n, ansatz_layers=0, ansatz_type=ansatz).to(DEV)
    assert bm.num_ansatz_params == 0
    np.testing.assert_allclose(bm.get_probabilities().detach().cpu().numpy(), expect, rtol=0, atol=1e-15)
