"""Host-side helpers with the reference's names and semantics (reference utils.py).

Only the two functions on the KSD hot path are mirrored:

* ``generate_all_binary_outcomes`` (utils.py:77-91) -- defines the index <-> bitstring
  order for the whole project: outcome i is ``bin(i).zfill(n)``, tuple position 0 is
  the MOST significant bit of i and is wire 0 of the circuit.
* ``calculate_tvd`` (utils.py:6-36) -- evaluation metric of the trainers.

Plotting (utils.py:38-67) is out of scope.
"""
import numpy as np


def generate_all_binary_outcomes(num_vars):
    """All binary outcomes for ``num_vars`` variables, lexicographic, MSB first (utils.py:77-91)."""
    if num_vars == 0:
        return [()]
    if num_vars == 1:
        return [(0,), (1,)]
    outcomes = []
    for i in range(2 ** num_vars):
        outcomes.append(tuple((i >> (num_vars - 1 - b)) & 1 for b in range(num_vars)))
    return outcomes


def outcome_index(z_tuple):
    """Inverse of ``generate_all_binary_outcomes``: tuple (MSB first) -> index."""
    idx = 0
    for b in z_tuple:
        idx = (idx << 1) | int(b)
    return idx


def calculate_tvd(p_true, p_approx):
    """Total variation distance between two discrete distributions (utils.py:6-36)."""
    if isinstance(p_true, dict) and isinstance(p_approx, dict):
        all_outcomes = set(p_true.keys()) | set(p_approx.keys())
        tvd = 0.0
        for outcome in all_outcomes:
            tvd += np.abs(p_true.get(outcome, 0.0) - p_approx.get(outcome, 0.0))
        return 0.5 * tvd
    elif isinstance(p_true, np.ndarray) and isinstance(p_approx, np.ndarray):
        if p_true.shape != p_approx.shape:
            raise ValueError("Probability arrays must have the same shape for simple TVD calculation.")
        return 0.5 * np.sum(np.abs(p_true - p_approx))
    else:
        raise TypeError("Inputs p_true and p_approx must be both dicts or both np.arrays.")
