"""Analytic known-answer tests for the oracle circuit.

The reference's circuit backend (PennyLane) is absent and untested in the reference, so
q_theta parity is "unpinned" by the reference; these KATs and the agreement of two
independent implementations are what pins the conventions (wire 0 = MSB, CNOT direction,
rotation signs, parameter order)."""
import math

import numpy as np
import pytest

from oracle import circuit as oc


@pytest.mark.parametrize("ansatz", ["hardware_efficient", "all_to_all"])
@pytest.mark.parametrize("n,L", [(1, 1), (2, 2), (3, 4), (5, 3)])
def test_theta_zero_uniform(ansatz, n, L):
    q = oc.probs(ansatz, n, L, np.zeros(oc.num_params(ansatz, n, L)))
    np.testing.assert_allclose(q, np.full(2 ** n, 2.0 ** -n), atol=1e-15)


@pytest.mark.parametrize("n,L", [(1, 1), (3, 2), (4, 3)])
def test_basic_theta_zero_delta(n, L):
    q = oc.probs("basic", n, L, np.zeros(oc.num_params("basic", n, L)))
    e = np.zeros(2 ** n); e[0] = 1
    np.testing.assert_allclose(q, e, atol=1e-15)


def test_n1_hardware_efficient_closed_form():
    a, b, c = 0.3, 0.9, -1.1
    q = oc.probs("hardware_efficient", 1, 1, [a, b, c])
    np.testing.assert_allclose(q, [(1 - math.sin(b)) / 2, (1 + math.sin(b)) / 2], atol=1e-15)
    np.testing.assert_allclose(q, [0.108336545186, 0.891663454814], atol=1e-11)
    q2 = oc.probs("hardware_efficient", 1, 1, [2.2, b, 0.4])
    np.testing.assert_allclose(q2, q, atol=1e-15)      # independent of a and c


def test_n2_basic_pins_cnot_direction_and_msb():
    t = [0.4, 0.2, 1.3, 0.5]
    q = oc.probs("basic", 2, 1, t)
    c0, s0 = math.cos(t[0] / 2) ** 2, math.sin(t[0] / 2) ** 2
    c1, s1 = math.cos(t[2] / 2) ** 2, math.sin(t[2] / 2) ** 2
    np.testing.assert_allclose(q, [c0 * c1, c0 * s1, s0 * s1, s0 * c1], atol=1e-15)
    np.testing.assert_allclose(q, [0.608735639904, 0.351794857098, 0.014455728590, 0.025013774409], atol=1e-11)


@pytest.mark.parametrize("ansatz", oc.ANSATZ_TYPES)
@pytest.mark.parametrize("n,L", [(2, 1), (3, 4), (4, 2), (6, 2)])
def test_two_implementations_agree(ansatz, n, L):
    rng = np.random.default_rng(n * 10 + L)
    th = rng.uniform(-np.pi, np.pi, oc.num_params(ansatz, n, L))
    q1 = oc.probs(ansatz, n, L, th)
    q2 = oc.probs(ansatz, n, L, th, dense=True)
    np.testing.assert_allclose(q1, q2, atol=5e-15)
    assert abs(q1.sum() - 1) < 1e-14


@pytest.mark.parametrize("ansatz", oc.ANSATZ_TYPES)
def test_gate_and_param_counts(ansatz):
    for n, L in [(3, 4), (8, 4), (16, 6)]:
        g = oc.gate_list(ansatz, n, L)
        P = oc.num_params(ansatz, n, L)
        assert sum(1 for x in g if x[2] is not None) == P
        if ansatz == "hardware_efficient":
            assert P == 3 * n * L
            assert len(g) == n + 4 * n * L + ((L + 1) // 2) * ((n - 2 + 1) // 2)
        elif ansatz == "all_to_all":
            assert len(g) == n + L * (3 * n + n * (n - 1) // 2)
        else:
            assert P == 2 * n * L and len(g) == L * 3 * n
    if ansatz == "hardware_efficient":
        assert len(oc.gate_list(ansatz, 16, 6)) == 421 and len(oc.gate_list(ansatz, 20, 8)) == 696


@pytest.mark.parametrize("ansatz", oc.ANSATZ_TYPES)
def test_param_shift_equals_finite_difference(ansatz):
    n, L = 4, 2
    rng = np.random.default_rng(3)
    th = rng.uniform(-1, 1, oc.num_params(ansatz, n, L))
    w = rng.normal(size=2 ** n)
    g = oc.paramshift_vjp(ansatz, n, L, th, w)
    h = 1e-6
    fd = np.zeros_like(th)
    for p in range(th.size):
        tp = th.copy(); tp[p] += h
        tm = th.copy(); tm[p] -= h
        fd[p] = (w @ oc.probs(ansatz, n, L, tp) - w @ oc.probs(ansatz, n, L, tm)) / (2 * h)
    np.testing.assert_allclose(g, fd, atol=2e-9)


def test_batched_simulator_matches_single():
    rng = np.random.default_rng(0)
    for ansatz in oc.ANSATZ_TYPES:
        n, L = 5, 2
        th = rng.uniform(-2, 2, (3, oc.num_params(ansatz, n, L)))
        qb = oc.probs_batched(ansatz, n, L, th)
        for b in range(3):
            np.testing.assert_allclose(qb[b], oc.probs(ansatz, n, L, th[b]), atol=1e-15)
