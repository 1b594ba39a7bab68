"""NumPy fp64 restatement of the reference's Stein-kernel maths (test infrastructure only).

Each function cites the reference lines it follows.  PINNED: checked against the
known answers in stein_utils.py:205-251 and the golden vectors under
tests/golden/ (captured by importing the reference, see make_golden.py).

Outcome order everywhere = utils.generate_all_binary_outcomes (utils.py:77-91):
index i <-> bin(i).zfill(n), tuple position 0 = MSB of i.
"""
import math
import numpy as np


# --------------------------------------------------------------------------- utils.py
def generate_all_binary_outcomes(n):
    """utils.py:77-91."""
    if n == 0:
        return [()]
    return [tuple((i >> (n - 1 - b)) & 1 for b in range(n)) for i in range(2 ** n)]


def calculate_tvd(p_true, p_approx):
    """utils.py:6-36."""
    if isinstance(p_true, dict) and isinstance(p_approx, dict):
        keys = set(p_true) | set(p_approx)
        return 0.5 * sum(abs(p_true.get(k, 0.0) - p_approx.get(k, 0.0)) for k in keys)
    if isinstance(p_true, np.ndarray) and isinstance(p_approx, np.ndarray):
        if p_true.shape != p_approx.shape:
            raise ValueError("Probability arrays must have the same shape for simple TVD calculation.")
        return 0.5 * float(np.sum(np.abs(p_true - p_approx)))
    raise TypeError("Inputs p_true and p_approx must be both dicts or both np.arrays.")


# --------------------------------------------------------------------------- stein_utils.py
def flip_bit(z, index):
    """stein_utils.py:7-11."""
    z = list(z)
    z[index] = 1 - z[index]
    return tuple(z)


def hamming_distance(z1, z2):
    """stein_utils.py:13-28."""
    return float(np.sum(np.abs(np.asarray(z1, np.float64) - np.asarray(z2, np.float64))))


def base_hamming_kernel(z1, z2, num_vars, length_scale=1.0):
    """stein_utils.py:30-55: exp(-||z1-z2||_1 / (num_vars*length_scale))."""
    if num_vars == 0:
        return 1.0
    d = hamming_distance(z1, z2)
    denom = float(num_vars) * float(length_scale)
    if denom == 0:
        return 1.0 if d == 0 else 0.0
    return math.exp(-d / denom)


def joint_probability(bn, full_assignment):
    """bayesian_network.py:111-146 for dict CPTs (callables evaluated the same way)."""
    if len(full_assignment) != len(bn.nodes):
        raise ValueError("Full assignment tuple length must match the number of nodes.")
    assign = {name: full_assignment[i] for i, name in enumerate(bn.nodes)}
    p = 1.0
    for name in bn.nodes:
        pv = tuple(assign[q] for q in bn.parents[name]) if name in bn.parents else ()
        cpt = bn.cpts[name]
        pd = cpt(pv) if callable(cpt) else cpt.get(pv)
        if pd is None:
            raise ValueError(f"CPT entry for node {name} with parent values {pv} not found.")
        p *= pd[assign[name]]
    return p


def compute_prob_joint_xz(bn, x_dict, z, latent_names, observed_names=None):
    """stein_utils.py:58-112: p(x, z), marginalising every other BN node."""
    cur = dict(x_dict) if x_dict else {}
    for i, name in enumerate(latent_names):
        cur[name] = z[i]
    others = [nd for nd in bn.nodes if nd not in cur]
    if not others:
        return float(joint_probability(bn, tuple(cur[nd] for nd in bn.nodes)))
    tot = 0.0
    for oa in generate_all_binary_outcomes(len(others)):
        full = dict(cur)
        for i, nm in enumerate(others):
            full[nm] = oa[i]
        tot += joint_probability(bn, tuple(full[nd] for nd in bn.nodes))
    return float(tot)


def score_for_z(bn, x_dict, z, latent_names, observed_names=None):
    """stein_utils.py:115-136: s_i = 1 - p(x, flip_i z)/p(x, z); zeros if |p| < 1e-12."""
    n = len(latent_names)
    s = np.zeros(n, dtype=np.float64)
    p = compute_prob_joint_xz(bn, x_dict, z, latent_names, observed_names)
    if abs(p) < 1e-12:
        return s
    for i in range(n):
        s[i] = 1.0 - (compute_prob_joint_xz(bn, x_dict, flip_bit(z, i), latent_names, observed_names) / p)
    return s


def score_matrix(bn, x_dict, latent_names, observed_names=None):
    """ksd_vi_quantum.py:70-75 (_precompute_all_s_p): S[z_index, b], shape [2^n, n]."""
    n = len(latent_names)
    return np.stack([score_for_z(bn, x_dict, z, latent_names, observed_names)
                     for z in generate_all_binary_outcomes(n)]) if n > 0 else np.zeros((1, 0))


def joint_vector(bn, x_dict, latent_names):
    n = len(latent_names)
    return np.array([compute_prob_joint_xz(bn, x_dict, z, latent_names)
                     for z in generate_all_binary_outcomes(n)], dtype=np.float64)


def stein_kernel_value(z1, z2, s1, s2, num_vars, length_scale=1.0):
    """stein_utils.py:138-197 (Eq. 13): term-by-term restatement."""
    n = num_vars
    if n == 0:
        return 0.0
    kf = lambda a, b: base_hamming_kernel(a, b, n, length_scale)
    k = kf(z1, z2)
    term1 = float(np.dot(s1, s2)) * k
    d2 = np.array([k - kf(z1, flip_bit(z2, j)) for j in range(n)])
    term2 = -float(np.dot(s1, d2))
    d1 = np.array([k - kf(flip_bit(z1, i), z2) for i in range(n)])
    term3 = -float(np.dot(d1, s2))
    tr = 0.0
    for i in range(n):
        z1n, z2n = flip_bit(z1, i), flip_bit(z2, i)
        tr += k - kf(z1, z2n) - kf(z1n, z2) + kf(z1n, z2n)
    return term1 + term2 + term3 + tr


def gram_loop(S, n, length_scale=1.0):
    """K_p[i, j] by the reference's double loop (ksd_vi_quantum.py:125-141); n <= ~6."""
    outs = generate_all_binary_outcomes(n)
    N = len(outs)
    K = np.zeros((N, N))
    for i in range(N):
        for j in range(N):
            K[i, j] = stein_kernel_value(outs[i], outs[j], S[i], S[j], n, length_scale)
    return K


# --------------------------------------------------------------------------- closed forms
def gram_closed_form(S, n, length_scale=1.0):
    """Vectorised K_p, algebraically equal to ``gram_loop``:

        k_p(i,j) = a^d * sum_b [ S_ib S_jb - c_b (S_ib + S_jb) + 2 c_b ],
        a = exp(-1/(n l)), d = popcount(i xor j),
        c_b = 1-a if bit b of i and j agree else 1-1/a.
    """
    N = 2 ** n
    a = math.exp(-1.0 / (n * length_scale))
    idx = np.arange(N)
    x = idx[:, None] ^ idx[None, :]
    bits = ((x[:, :, None] >> (n - 1 - np.arange(n))[None, None, :]) & 1).astype(np.float64)
    d = bits.sum(-1)
    c = np.where(bits > 0, 1.0 - 1.0 / a, 1.0 - a)
    T = S[:, None, :] * S[None, :, :] - c * (S[:, None, :] + S[None, :, :]) + 2.0 * c
    return (a ** d) * T.sum(-1)


def kbase_apply(v, n, a):
    """K_base v with K_base = M^{(x) n}, M = [[1, a], [a, 1]]: n butterfly passes."""
    v = v.reshape((2,) * n).astype(np.float64)
    for ax in range(n):
        v0 = np.take(v, 0, axis=ax)
        v1 = np.take(v, 1, axis=ax)
        v = np.stack([v0 + a * v1, a * v0 + v1], axis=ax)
    return v.reshape(-1)


def flip_axis(v, n, b):
    """(flip_b v)[i] = v[i with outcome-bit b flipped]; bit b = tuple position b (MSB first)."""
    return np.flip(v.reshape((2,) * n), axis=b).reshape(-1)


def stein_matvec_kron(S, q, n, length_scale=1.0):
    """y = K_p q matrix-free, O(n^2 2^n)."""
    a = math.exp(-1.0 / (n * length_scale))
    u = kbase_apply(q, n, a)
    y = np.zeros_like(u)
    for b in range(n):
        sb = S[:, b]
        w = kbase_apply(sb * q, n, a)
        du = u - flip_axis(u, n, b)
        dw = w - flip_axis(w, n, b)
        y += sb * w - sb * du - dw + 2.0 * du
    return y
