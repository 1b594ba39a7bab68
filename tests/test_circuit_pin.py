"""Independent pin of the circuit half of the oracle (CPU only).

The reference's circuit arithmetic lives in PennyLane, which is absent here and has no fixture in the reference:
q_theta stays "parity unpinned" BY THE REFERENCE.  What this file adds is evidence held inside the repo that the
oracle (oracle/circuit.py, oracle/cpu_port.c) implements the published operator definitions and the gate order of
quantum_born_machine.py:57-128, from code that shares NOTHING with oracle/circuit.py:

  * gates defined as matrix exponentials, RX/RY/RZ(t) = scipy.linalg.expm(-i t P / 2) -- pins the rotation signs;
  * a flat-index simulator (loops over basis-state integers, no tensordot / reshape, its own gate sequence);
  * a sympy-symbolic simulation of n = 2 `hardware_efficient`, L = 2, that depends on the RX and RZ angles
    (at L = 1 the distribution provably does not: RX acts on |+>, RZ is followed only by a CNOT);
  * an n = 3 case that is sensitive to CZ(0, 2), to the direction of the wrap-around CNOT(2, 0) and to the sign
    conventions of RX and RZ: the oracle agrees with the definition and visibly disagrees with each variant.
"""
import numpy as np
import pytest
import scipy.linalg as sla

from oracle import circuit as oc

PX = np.array([[0, 1], [1, 0]], dtype=complex)
PY = np.array([[0, -1j], [1j, 0]], dtype=complex)
PZ = np.array([[1, 0], [0, -1]], dtype=complex)
HAD = (PX + PZ) / np.sqrt(2.0)


def rot(P, t, sign=-1.0):
    return sla.expm(sign * 0.5j * t * P)          # R_P(t) = exp(-i t P / 2)


def program(ansatz, n, L, variant=None):
    """Operations in program order as ('u', wire, pauli | 'H', param index | None) / ('cx', c, t) / ('cz', a, b).
    Written from quantum_born_machine.py:58-87 (hardware_efficient), :90-111 (all_to_all), :114-128 (basic).
    variant: None | 'no_cz02' | 'wrap_reversed' (deliberately WRONG circuits for the sensitivity checks)."""
    ops, p = [], 0
    rots = {"hardware_efficient": "XYZ", "all_to_all": "XYZ", "basic": "YZ"}[ansatz]
    if ansatz != "basic":
        ops += [("u", w, "H", None) for w in range(n)]
    for layer in range(L):
        for w in range(n):
            for ax in rots:
                ops.append(("u", w, ax, p)); p += 1
        if n < 2:
            continue
        if ansatz == "all_to_all":
            ops += [("cz", a, b) for a in range(n) for b in range(a + 1, n)]
            continue
        ops += [("cx", w, w + 1) for w in range(n - 1)]
        if n > 2:
            ops.append(("cx", 0, n - 1) if variant == "wrap_reversed" else ("cx", n - 1, 0))
        if ansatz == "hardware_efficient" and layer % 2 == 0 and n > 2:
            for a in range(0, n - 2, 2):
                if not (variant == "no_cz02" and (a, a + 2) == (0, 2)):
                    ops.append(("cz", a, a + 2))
    return ops, p


def flat_probs(ansatz, n, L, theta, variant=None, rx_sign=-1.0, rz_sign=-1.0):
    """Born probabilities by looping over basis-state integers; wire w <-> bit n-1-w (wire 0 = MSB, qml.probs order)."""
    ops, P = program(ansatz, n, L, variant)
    assert P == len(theta)
    N = 1 << n
    psi = np.zeros(N, dtype=complex)
    psi[0] = 1.0
    for op in ops:
        if op[0] == "u":
            _, w, ax, pi = op
            U = HAD if ax == "H" else rot({"X": PX, "Y": PY, "Z": PZ}[ax], theta[pi],
                                          {"X": rx_sign, "Y": -1.0, "Z": rz_sign}[ax])
            m = 1 << (n - 1 - w)
            new = psi.copy()
            for i in range(N):
                if not i & m:
                    a, b = psi[i], psi[i | m]
                    new[i] = U[0, 0] * a + U[0, 1] * b
                    new[i | m] = U[1, 0] * a + U[1, 1] * b
            psi = new
        elif op[0] == "cx":
            cm, tm = 1 << (n - 1 - op[1]), 1 << (n - 1 - op[2])
            new = np.empty_like(psi)
            for i in range(N):
                new[i ^ tm if i & cm else i] = psi[i]
            psi = new
        else:
            am, bm = 1 << (n - 1 - op[1]), 1 << (n - 1 - op[2])
            for i in range(N):
                if (i & am) and (i & bm):
                    psi[i] = -psi[i]
    return (psi * psi.conj()).real


@pytest.mark.parametrize("ansatz", oc.ANSATZ_TYPES)
@pytest.mark.parametrize("n,L", [(1, 2), (2, 2), (3, 2), (4, 3), (5, 2), (6, 2)])
def test_oracle_equals_flat_expm_simulator(ansatz, n, L):
    rng = np.random.default_rng(1000 * n + 10 * L + len(ansatz))
    th = rng.uniform(-np.pi, np.pi, oc.num_params(ansatz, n, L))
    q = flat_probs(ansatz, n, L, th)
    np.testing.assert_allclose(oc.probs(ansatz, n, L, th), q, rtol=0, atol=2e-15)
    np.testing.assert_allclose(oc.probs(ansatz, n, L, th, dense=True), q, rtol=0, atol=5e-15)
    from oracle import cpu_port as cp
    if cp.available():
        np.testing.assert_allclose(cp.circuit_probs(ansatz, n, L, th)[0], q, rtol=0, atol=2e-15)
    assert len(program(ansatz, n, L)[0]) == len(oc.gate_list(ansatz, n, L))


def test_elementary_matrices_are_the_matrix_exponentials():
    for t in (0.0, 0.37, -2.9, np.pi / 2, -np.pi / 2):
        np.testing.assert_allclose(oc.matrix_1q("RX", t), rot(PX, t), atol=1e-15)
        np.testing.assert_allclose(oc.matrix_1q("RY", t), rot(PY, t), atol=1e-15)
        np.testing.assert_allclose(oc.matrix_1q("RZ", t), rot(PZ, t), atol=1e-15)
    np.testing.assert_allclose(oc.matrix_1q("H"), HAD, atol=1e-15)


def test_sympy_symbolic_n2_hardware_efficient_L2_depends_on_rx_and_rz():
    """Exact symbolic simulation (sympy Matrix.exp of -i t P / 2, Kronecker products, CNOT as a permutation matrix) of
    n = 2 `hardware_efficient` L = 2; evaluated at random angles it equals the oracle, and the expression really
    depends on RZ angles of the first layer and RX angles of the second (the pin the theta-independent KATs lack)."""
    sp = pytest.importorskip("sympy")
    from sympy.physics.quantum import TensorProduct as TP
    th = sp.symbols("t0:12", real=True)
    I2 = sp.eye(2)
    sx, sy, sz = sp.Matrix([[0, 1], [1, 0]]), sp.Matrix([[0, -sp.I], [sp.I, 0]]), sp.Matrix([[1, 0], [0, -1]])
    R = lambda P, t: (-sp.I * t / 2 * P).exp()
    Hm = sp.Matrix([[1, 1], [1, -1]]) / sp.sqrt(2)
    CX01 = sp.Matrix([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0]])      # control = wire 0 = MSB
    psi = TP(Hm, Hm) * sp.Matrix([1, 0, 0, 0])
    p = 0
    for layer in range(2):
        for w in range(2):
            for P in (sx, sy, sz):
                U = R(P, th[p]); p += 1
                psi = (TP(U, I2) if w == 0 else TP(I2, U)) * psi
        psi = CX01 * psi
    probs = [sp.re(sp.expand_complex(a * sp.conjugate(a))) for a in psi]
    f = sp.lambdify(th, probs, "numpy")
    rng = np.random.default_rng(2)
    for _ in range(3):
        v = rng.uniform(-np.pi, np.pi, 12)
        np.testing.assert_allclose(np.array(f(*v), dtype=float), oc.probs("hardware_efficient", 2, 2, v), rtol=0, atol=1e-14)
    v = rng.uniform(0.3, 1.2, 12)
    base = np.array(f(*v), dtype=float)
    for k in (2, 5, 6, 9):            # RZ angles of layer 0 and RX angles of layer 1, both wires (layer 0's RX acts on
                                      # |+>, an X eigenstate: a global phase, so those angles cannot matter)
        assert any(pr.has(th[k]) for pr in probs)
        w = v.copy(); w[k] += 0.7
        assert np.abs(np.array(f(*w), dtype=float) - base).max() > 1e-3
        # ... and flipping that rotation's sign convention would be seen
        w = v.copy(); w[k] = -w[k]
        assert np.abs(np.array(f(*w), dtype=float) - base).max() > 1e-3


def test_n3_case_is_sensitive_to_cz02_wrap_cnot_and_rotation_signs():
    n, L = 3, 2
    th = np.random.default_rng(33).uniform(0.4, 2.6, oc.num_params("hardware_efficient", n, L))
    right = flat_probs("hardware_efficient", n, L, th)
    got = oc.probs("hardware_efficient", n, L, th)
    np.testing.assert_allclose(got, right, rtol=0, atol=2e-15)
    wrong = {
        "CZ(0,2) dropped": flat_probs("hardware_efficient", n, L, th, variant="no_cz02"),
        "CNOT(2,0) reversed": flat_probs("hardware_efficient", n, L, th, variant="wrap_reversed"),
        "RX = exp(+i t X/2)": flat_probs("hardware_efficient", n, L, th, rx_sign=+1.0),
        "RZ = exp(+i t Z/2)": flat_probs("hardware_efficient", n, L, th, rz_sign=+1.0),
    }
    for name, q in wrong.items():
        assert abs(q.sum() - 1) < 1e-14
        assert np.abs(q - got).max() > 1e-3, f"the case cannot tell: {name}"
    # `basic` (no Hadamards, RY/RZ only): CNOT direction of the ring
    thb = np.random.default_rng(34).uniform(0.4, 2.6, oc.num_params("basic", 3, 2))
    gb = oc.probs("basic", 3, 2, thb)
    np.testing.assert_allclose(gb, flat_probs("basic", 3, 2, thb), rtol=0, atol=2e-15)
    assert np.abs(flat_probs("basic", 3, 2, thb, variant="wrap_reversed") - gb).max() > 1e-3
    # (`basic` cannot see the RZ sign: RY and CNOT are real, so RZ(t) -> RZ(-t) conjugates the whole state)
    np.testing.assert_allclose(flat_probs("basic", 3, 2, thb, rz_sign=+1.0), gb, rtol=0, atol=1e-15)
