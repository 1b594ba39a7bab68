"""Pin the CPU oracle's Stein half against the reference's own known answers
(stein_utils.py:205-251) and the golden vectors captured from the reference import."""
import math

import numpy as np
import pytest

from oracle import stein as os_
from oracle import ksd as ok
from tensornetworks_amd.bayesian_network import (BayesianNetwork, get_sprinkler_network,
                                                  synthetic_network)
from conftest import golden


def two_node():
    bn = BayesianNetwork()
    bn.add_node('A', cpt={(): {0: 0.8, 1: 0.2}})
    bn.add_node('B', cpt={(0,): {0: 0.7, 1: 0.3}, (1,): {0: 0.4, 1: 0.6}}, parent_names=['A'])
    return bn


# --- the seven assertions of stein_utils.py:205-251, restated -------------------------------
def test_reference_known_answers():
    assert os_.flip_bit((0, 0, 0), 0) == (1, 0, 0)                                   # :205
    assert os_.hamming_distance([0, 0, 1, 1], [1, 0, 0, 1]) == 2.0                   # :208-211
    assert math.isclose(os_.base_hamming_kernel([0, 0, 1, 1], [1, 0, 0, 1], 4, 1.0),
                        math.exp(-2.0 / 4))                                          # :214-217
    bn = two_node()
    x = {'B': 1}
    assert np.isclose(os_.compute_prob_joint_xz(bn, x, (1,), ['A'], ['B']), 0.12)    # :229-230
    s1 = os_.score_for_z(bn, x, (1,), ['A'], ['B'])
    s0 = os_.score_for_z(bn, x, (0,), ['A'], ['B'])
    assert np.isclose(s1[0], -1.0) and np.isclose(s0[0], 0.5)                        # :233-236
    assert np.isclose(os_.stein_kernel_value((0,), (1,), s0, s1, 1, 1.0), 2 * math.exp(-1.0) - 2.5)   # :245-247
    assert np.isclose(os_.stein_kernel_value((0,), (0,), s0, s0, 1, 1.0), 1.25 - math.exp(-1.0))      # :249-251


def test_kernel_edge_cases():
    assert os_.base_hamming_kernel([], [], 0) == 1.0                                 # :36-40
    assert os_.base_hamming_kernel([0, 1], [0, 1], 2, 0.0) == 1.0                    # :52-54
    assert os_.base_hamming_kernel([0, 1], [1, 1], 2, 0.0) == 0.0
    assert os_.stein_kernel_value((), (), np.zeros(0), np.zeros(0), 0) == 0.0        # :151-152
    assert os_.generate_all_binary_outcomes(0) == [()]
    assert os_.generate_all_binary_outcomes(2) == [(0, 0), (0, 1), (1, 0), (1, 1)]   # utils.py:80


def test_tvd():
    p1 = {'00': 0.25, '01': 0.25, '10': 0.25, '11': 0.25}
    p2 = {'00': 0.5, '01': 0.1, '10': 0.1, '11': 0.3}
    assert math.isclose(os_.calculate_tvd(p1, p2), 0.3)                              # utils.py:96-102
    assert math.isclose(os_.calculate_tvd(np.array([.25] * 4), np.array([.5, .1, .1, .3])), 0.3)
    with pytest.raises(TypeError):
        os_.calculate_tvd(p1, np.array([1.0]))


# --- golden vectors ---------------------------------------------------------------------------
@pytest.mark.parametrize("tag,w", [("w1", 1), ("w0", 0)])
def test_sprinkler_golden(tag, w):
    g = golden(f"sprinkler_{tag}.npz")
    bn = get_sprinkler_network(False)
    lat, obs, x = ['C', 'S', 'R'], ['W'], {'W': w}
    S = os_.score_matrix(bn, x, lat, obs)
    np.testing.assert_array_equal(S, g["S"])                 # same fp64 operations -> bit-exact
    np.testing.assert_array_equal(os_.joint_vector(bn, x, lat), g["pxz"])
    K = os_.gram_loop(S, 3)
    np.testing.assert_allclose(K, g["K"], rtol=1e-13, atol=1e-13 * np.abs(g["K"]).max())
    Kc = os_.gram_closed_form(S, 3)
    np.testing.assert_allclose(Kc, g["K"], rtol=0, atol=2e-15 * np.abs(g["K"]).max())
    post, p_obs = bn.get_true_posterior(lat, x)
    assert p_obs == float(g["p_observed"])
    np.testing.assert_array_equal(np.array([post[z] for z in os_.generate_all_binary_outcomes(3)]),
                                  g["posterior"])
    for nm in ("rand", "uniform"):
        q = g[f"q_{nm}"]
        assert math.isclose(ok.ksd_loss(Kc, q), float(g[f"loss_{nm}"]), rel_tol=1e-12)
        np.testing.assert_allclose(ok.ksd_grad_q(Kc, q), g[f"dLdq_{nm}"], rtol=1e-10,
                                   atol=1e-12 * np.abs(g[f"dLdq_{nm}"]).max())
        y = os_.stein_matvec_kron(S, q, 3)
        np.testing.assert_allclose(y, g["K"] @ q, rtol=0, atol=1e-13 * np.abs(g["K"]).max())
    # at the true posterior the quadratic form cancels to ~1e-16 of its terms
    qp = g["posterior"]
    scale = float(np.abs(qp[:, None] * qp[None, :] * g["K"]).sum())
    assert abs(ok.ksd_squared(Kc, qp) - float(g["ksd2_posterior"])) < 1e-13 * scale


def test_sprinkler_headline_numbers():
    g = golden("sprinkler_w1.npz")
    assert math.isclose(float(g["p_observed"]), 0.65, rel_tol=1e-12)
    assert math.isclose(float(g["loss_uniform"]), 48.62993563065114, rel_tol=1e-12)
    np.testing.assert_allclose(g["pxz"], [0.002, 0.045, 0.18, 0.0495, 0.0009, 0.324, 0.009, 0.0396], rtol=1e-12)


@pytest.mark.parametrize("sd", [0, 1, 2])
def test_sprinkler_random_cpts_golden(sd):
    g = golden(f"sprinkler_rand{sd}.npz")
    np.random.seed(sd)
    bn = get_sprinkler_network(True)            # same RNG stream and draw order as the reference
    S = os_.score_matrix(bn, {'W': 1}, ['C', 'S', 'R'], ['W'])
    np.testing.assert_array_equal(S, g["S"])
    np.testing.assert_allclose(os_.gram_closed_form(S, 3), g["K"], rtol=0, atol=3e-15 * np.abs(g["K"]).max())


@pytest.mark.parametrize("name", ["synthetic_n4_s0", "synthetic_n5_s0", "synthetic_n5_s1",
                                  "synthetic_n6_s0", "synthetic_n8_s0"])
def test_synthetic_golden(name):
    g = golden(name + ".npz")
    n, seed = int(g["n"][0]), int(g["seed"][0])
    bn, lat, obs, x = synthetic_network(n, seed)
    S = os_.score_matrix(bn, x, lat, obs)
    np.testing.assert_array_equal(S, g["S"])
    np.testing.assert_array_equal(os_.joint_vector(bn, x, lat), g["pxz"])
    Kc = os_.gram_closed_form(S, n)[g["rows"]]
    np.testing.assert_allclose(Kc, g["K"], rtol=0, atol=3e-15 * np.abs(g["K"]).max())
    if n <= 5:
        np.testing.assert_allclose(os_.gram_loop(S, n), g["K"], rtol=1e-13, atol=1e-13 * np.abs(g["K"]).max())
    rng = np.random.default_rng(5)
    q = rng.random(2 ** n); q /= q.sum()
    Kfull = os_.gram_closed_form(S, n)
    y = os_.stein_matvec_kron(S, q, n)
    np.testing.assert_allclose(y, Kfull @ q, rtol=0, atol=1e-13 * np.abs(Kfull).max())


def test_two_node_golden():
    g = golden("two_node.npz")
    bn = two_node()
    S = os_.score_matrix(bn, {'B': 1}, ['A'], ['B'])
    np.testing.assert_array_equal(S, g["S"])
    np.testing.assert_allclose(os_.gram_closed_form(S, 1), g["K"], rtol=1e-14)


def test_classical_trainer_trace():
    """The KSD loop shared by ksd_vi.py:114-134 and ksd_vi_quantum.py:123-145:
    for the q each epoch saw, our KSD equals the loss the reference recorded."""
    g = golden("classical_trace.npz")
    sp = golden("sprinkler_w1.npz")
    K = os_.gram_closed_form(sp["S"], 3)
    q_loss = g["q_all"][::2]                     # every epoch makes two forwards: loss, entropy
    for q, l in zip(q_loss, g["loss_ksd"]):
        assert math.isclose(ok.ksd_loss(K, q), float(l), rel_tol=1e-12)


def test_gram_properties():
    g = golden("synthetic_n6_s0.npz")
    K = os_.gram_closed_form(g["S"], 6)
    assert np.abs(K - K.T).max() <= 1e-12 * np.abs(K).max()
    ev = np.linalg.eigvalsh(0.5 * (K + K.T))
    assert ev.min() > -1e-9 * ev.max()                       # PSD
    p = g["pxz"] / g["pxz"].sum()
    assert np.abs(K @ p).max() < 1e-9 * np.abs(K).max()      # posterior in the null space
