"""NumPy fp64 restatement of the reference's parameterised circuits (test infrastructure only).

Follows quantum_born_machine.py:57-128 (gate order, parameter consumption) and
PennyLane's published operator definitions:

    H  = 1/sqrt(2) [[1, 1], [1, -1]]
    RX(t) = exp(-i t X / 2) = [[cos t/2, -i sin t/2], [-i sin t/2, cos t/2]]
    RY(t) = exp(-i t Y / 2) = [[cos t/2, -sin t/2], [sin t/2, cos t/2]]
    RZ(t) = diag(exp(-i t/2), exp(+i t/2))
    CNOT(wires=[control, target]),  CZ(wires=[a, b])

Initial state |0...0>; ``qml.probs(wires=range(n))`` returns probabilities in
lexicographic order with wire 0 the MOST significant bit, which coincides with
``utils.generate_all_binary_outcomes`` (utils.py:77-91), as
quantum_born_machine.py:143-150 assumes.

PARITY UNPINNED by the reference (PennyLane absent, no fixtures): see
``oracle/__init__.py``.  Two independent implementations live here
(``simulate`` = gate-by-gate tensordot, ``simulate_dense`` = dense Kronecker
unitaries) and must agree; analytic known answers are in tests/test_oracle_circuit.py.
"""
import numpy as np

ANSATZ_TYPES = ("hardware_efficient", "all_to_all", "basic")


def num_params(ansatz_type, n, layers):
    """quantum_born_machine.py:31-38."""
    if ansatz_type in ("hardware_efficient", "all_to_all"):
        return layers * 3 * n
    return layers * 2 * n


def gate_list(ansatz_type, n, layers):
    """Gate sequence as (kind, wires, param_index|None), program order.

    hardware_efficient: quantum_born_machine.py:58-87
    all_to_all:         quantum_born_machine.py:90-111
    basic:              quantum_born_machine.py:114-128
    """
    g = []
    p = 0
    if ansatz_type == "hardware_efficient":
        for i in range(n):
            g.append(("H", (i,), None))
        for layer in range(layers):
            for i in range(n):
                g.append(("RX", (i,), p)); p += 1
                g.append(("RY", (i,), p)); p += 1
                g.append(("RZ", (i,), p)); p += 1
            if n > 1:
                for i in range(n - 1):
                    g.append(("CNOT", (i, i + 1), None))
                if n > 2:
                    g.append(("CNOT", (n - 1, 0), None))
                if layer % 2 == 0 and n > 2:
                    for i in range(0, n - 2, 2):
                        g.append(("CZ", (i, i + 2), None))
    elif ansatz_type == "all_to_all":
        for i in range(n):
            g.append(("H", (i,), None))
        for layer in range(layers):
            for i in range(n):
                g.append(("RX", (i,), p)); p += 1
                g.append(("RY", (i,), p)); p += 1
                g.append(("RZ", (i,), p)); p += 1
            if n > 1:
                for i in range(n):
                    for j in range(i + 1, n):
                        g.append(("CZ", (i, j), None))
    else:  # basic
        for layer in range(layers):
            for i in range(n):
                g.append(("RY", (i,), p)); p += 1
                g.append(("RZ", (i,), p)); p += 1
            if n > 1:
                for i in range(n - 1):
                    g.append(("CNOT", (i, i + 1), None))
                if n > 2:
                    g.append(("CNOT", (n - 1, 0), None))
    assert p == num_params(ansatz_type, n, layers)
    return g


def matrix_1q(kind, t=None):
    if kind == "H":
        return np.array([[1, 1], [1, -1]], dtype=np.complex128) / np.sqrt(2.0)
    c, s = np.cos(t / 2.0), np.sin(t / 2.0)
    if kind == "RX":
        return np.array([[c, -1j * s], [-1j * s, c]], dtype=np.complex128)
    if kind == "RY":
        return np.array([[c, -s], [s, c]], dtype=np.complex128)
    if kind == "RZ":
        return np.array([[np.exp(-0.5j * t), 0], [0, np.exp(0.5j * t)]], dtype=np.complex128)
    raise ValueError(kind)


def apply_1q(state, U, w):
    """state: ndarray shape [2]*n, axis w == wire w (wire 0 = MSB of the flat index)."""
    state = np.tensordot(U, state, axes=([1], [w]))
    return np.moveaxis(state, 0, w)


def apply_cnot(state, c, t):
    state = state.copy()
    idx1 = [slice(None)] * state.ndim
    idx1[c] = 1
    sub = state[tuple(idx1)]                 # control == 1 slice (view of the copy)
    ax = t if t < c else t - 1               # axis of the target inside the slice
    state[tuple(idx1)] = np.flip(sub, axis=ax)
    return state


def apply_cz(state, a, b):
    state = state.copy()
    idx = [slice(None)] * state.ndim
    idx[a] = 1
    idx[b] = 1
    state[tuple(idx)] *= -1.0
    return state


def simulate(gates, n, theta):
    """Gate-by-gate statevector simulation; returns the flat complex128 state [2^n]."""
    theta = np.asarray(theta, dtype=np.float64)
    state = np.zeros((2,) * n, dtype=np.complex128) if n > 0 else np.ones((), np.complex128)
    if n > 0:
        state[(0,) * n] = 1.0
    for kind, wires, p in gates:
        if kind in ("H", "RX", "RY", "RZ"):
            state = apply_1q(state, matrix_1q(kind, None if p is None else theta[p]), wires[0])
        elif kind == "CNOT":
            state = apply_cnot(state, wires[0], wires[1])
        elif kind == "CZ":
            state = apply_cz(state, wires[0], wires[1])
        else:
            raise ValueError(kind)
    return state.reshape(-1)


def _kron_all(mats):
    out = np.ones((1, 1), dtype=np.complex128)
    for m in mats:
        out = np.kron(out, m)
    return out


def _dense_gate(kind, wires, n, t):
    I2 = np.eye(2, dtype=np.complex128)
    P0 = np.array([[1, 0], [0, 0]], dtype=np.complex128)
    P1 = np.array([[0, 0], [0, 1]], dtype=np.complex128)
    X = np.array([[0, 1], [1, 0]], dtype=np.complex128)
    Z = np.array([[1, 0], [0, -1]], dtype=np.complex128)
    if kind in ("H", "RX", "RY", "RZ"):
        mats = [I2] * n
        mats[wires[0]] = matrix_1q(kind, t)
        return _kron_all(mats)           # wire 0 leftmost factor == MSB
    a, b = wires
    m0 = [I2] * n; m0[a] = P0
    m1 = [I2] * n; m1[a] = P1; m1[b] = X if kind == "CNOT" else Z
    return _kron_all(m0) + _kron_all(m1)


def simulate_dense(gates, n, theta):
    """Independent second implementation: full 2^n x 2^n unitaries (n <= ~8)."""
    theta = np.asarray(theta, dtype=np.float64)
    state = np.zeros(2 ** n, dtype=np.complex128)
    state[0] = 1.0
    for kind, wires, p in gates:
        state = _dense_gate(kind, wires, n, None if p is None else theta[p]) @ state
    return state


def probs(ansatz_type, n, layers, theta, dense=False):
    """q_theta over all 2^n outcomes (float64), lexicographic, wire 0 = MSB."""
    g = gate_list(ansatz_type, n, layers)
    psi = (simulate_dense if dense else simulate)(g, n, theta)
    return psi.real ** 2 + psi.imag ** 2


def paramshift_vjp(ansatz_type, n, layers, theta, dLdq):
    """grad_p = sum_z dLdq[z] * (q(theta + pi/2 e_p)[z] - q(theta - pi/2 e_p)[z]) / 2.

    What `diff_method="parameter-shift"` (quantum_born_machine.py:58,:90,:114)
    computes when `loss.backward()` reaches the QNode (ksd_vi_quantum.py:150):
    two-term rule, shift pi/2, coefficient 1/2, one pair of circuits per parameter.
    """
    theta = np.asarray(theta, dtype=np.float64)
    grad = np.zeros_like(theta)
    for p in range(theta.size):
        tp = theta.copy(); tp[p] += np.pi / 2
        tm = theta.copy(); tm[p] -= np.pi / 2
        qp = probs(ansatz_type, n, layers, tp)
        qm = probs(ansatz_type, n, layers, tm)
        grad[p] = 0.5 * np.dot(dLdq, qp - qm)
    return grad


# ---------------------------------------------------------------------------
# Batched simulator used by bench.py's cpu_baseline ("port") leg: the same gate
# list applied to B states at once so that NumPy works on large arrays.
# ---------------------------------------------------------------------------
def probs_batched(ansatz_type, n, layers, thetas):
    thetas = np.asarray(thetas, dtype=np.float64)
    B = thetas.shape[0]
    g = gate_list(ansatz_type, n, layers)
    state = np.zeros((B,) + (2,) * n, dtype=np.complex128)
    state[(slice(None),) + (0,) * n] = 1.0
    for kind, wires, p in g:
        if kind in ("H", "RX", "RY", "RZ"):
            w = wires[0] + 1
            if p is None:
                U = np.broadcast_to(matrix_1q(kind), (B, 2, 2))
            else:
                U = np.stack([matrix_1q(kind, t) for t in thetas[:, p]])
            s = np.moveaxis(state, w, 1)                      # [B, 2, ...]
            shp = s.shape
            s = np.matmul(U, s.reshape(B, 2, -1)).reshape(shp)
            state = np.moveaxis(s, 1, w)
        elif kind == "CNOT":
            c, t = wires[0] + 1, wires[1] + 1
            state = np.ascontiguousarray(state)
            idx = [slice(None)] * state.ndim
            idx[c] = 1
            sub = state[tuple(idx)]
            ax = t if t < c else t - 1
            state[tuple(idx)] = np.flip(sub, axis=ax).copy()
        else:
            a, b = wires[0] + 1, wires[1] + 1
            state = np.ascontiguousarray(state)
            idx = [slice(None)] * state.ndim
            idx[a] = 1; idx[b] = 1
            state[tuple(idx)] *= -1.0
    psi = state.reshape(B, -1)
    return psi.real ** 2 + psi.imag ** 2
