"""Binary Bayesian-network data model for the score builder (host side).

Mirrors the part of the reference's ``BayesianNetwork`` that the KSD hot path touches
(bayesian_network.py:6-51 data model, :111-146 ``get_joint_probability``) plus the
evaluation helper ``get_true_posterior`` (:148-253) and the Sprinkler factory (:312-383),
with the same attribute names (``nodes``, ``parents``, ``cpts``, ``node_to_index``) so that
a reference ``BayesianNetwork`` instance and one of these are interchangeable wherever this
package takes a network (duck typing).

New here: ``pack_network`` flattens the dict-of-dict CPTs into dense arrays for the HIP
score kernel (``bornvi_score_from_cpts``), and ``synthetic_network`` builds the n-node
benchmark networks of SURVEY.md section 8(d) -- the reference ships only Sprinkler.
"""
from collections import defaultdict

import numpy as np

from .utils import generate_all_binary_outcomes

MAX_PARENTS = 8


class BayesianNetwork:
    """Discrete BN over binary variables; nodes must be added parents-first."""

    def __init__(self):
        self.nodes = []
        self.parents = defaultdict(list)
        self.cpts = {}
        self.node_to_index = {}

    def add_node(self, name, cpt, parent_names=None):
        """cpt: {parent_values_tuple: {0: p0, 1: p1}} or a callable returning such a dict."""
        if name in self.node_to_index:
            raise ValueError(f"Node {name} already exists.")
        for pn in parent_names or []:
            if pn not in self.node_to_index:
                raise ValueError(f"Parent node {pn} for {name} not found. Add parents first.")
        self.node_to_index[name] = len(self.nodes)
        self.nodes.append(name)
        if parent_names:
            self.parents[name] = list(parent_names)
        self.cpts[name] = cpt

    def _cpt_row(self, name, parent_values):
        entry = self.cpts[name]
        row = entry(parent_values) if callable(entry) else entry.get(parent_values)
        if row is None:
            raise ValueError(f"CPT entry for node {name} with parent values {parent_values} not found.")
        return row

    def get_joint_probability(self, full_assignment_tuple):
        """Product of CPT entries in node order (bayesian_network.py:111-146)."""
        if len(full_assignment_tuple) != len(self.nodes):
            raise ValueError("Full assignment tuple length must match the number of nodes.")
        value = dict(zip(self.nodes, full_assignment_tuple))
        prob = 1.0
        for name in self.nodes:
            pv = tuple(value[p] for p in self.parents[name]) if name in self.parents else ()
            prob *= self._cpt_row(name, pv)[value[name]]
        return prob

    def get_true_posterior(self, latent_vars_names, observed_vars_dict):
        """P(latent | observed) by enumeration; returns ({tuple: prob}, P(observed))
        like bayesian_network.py:148-253 (other nodes are marginalised)."""
        for nm in latent_vars_names:
            if nm not in self.node_to_index:
                raise ValueError("One or more latent variable names not in the network.")
        for nm in observed_vars_dict:
            if nm not in self.node_to_index:
                raise ValueError("One or more observed variable names not in the network.")
        if set(latent_vars_names) & set(observed_vars_dict):
            raise ValueError("Latent and observed variables must be disjoint.")
        joint = joint_table(self, latent_vars_names, observed_vars_dict)
        outs = generate_all_binary_outcomes(len(latent_vars_names))
        p_obs = float(sum(joint.tolist()))
        if p_obs == 0:
            print(f"Warning: P(Observed) is zero for evidence {observed_vars_dict}. Posterior is ill-defined.")
            return {z: 0.0 for z in outs}, p_obs
        return {z: float(v) / p_obs for z, v in zip(outs, joint)}, p_obs


def joint_table(bn, latent_names, x_dict):
    """p(x, z) for every latent outcome z (lexicographic), other nodes summed out; fp64, host."""
    others = [nd for nd in bn.nodes if nd not in latent_names and nd not in (x_dict or {})]
    outs = generate_all_binary_outcomes(len(latent_names))
    table = np.zeros(len(outs))
    for zi, z in enumerate(outs):
        cur = dict(x_dict or {})
        cur.update(zip(latent_names, z))
        tot = 0.0
        for oa in generate_all_binary_outcomes(len(others)):
            cur.update(zip(others, oa))
            tot += bn.get_joint_probability(tuple(cur[nd] for nd in bn.nodes))
        table[zi] = tot
    return table


def get_sprinkler_network(random_cpts=False):
    """Cloudy -> {Sprinkler, Rain} -> WetGrass with the reference's tables
    (bayesian_network.py:358-381) or U(0.01, 0.99) entries drawn from the global NumPy RNG
    in the reference's order (bayesian_network.py:321-356)."""
    if random_cpts:
        draw = lambda: np.random.uniform(0.01, 0.99)
        pc = draw()
        ps = [draw(), draw()]
        pr = [draw(), draw()]
        pw = [draw(), draw(), draw(), draw()]
        row = lambda p: {0: 1 - p, 1: p}
        c_tab = {(): row(pc)}
        s_tab = {(0,): row(ps[0]), (1,): row(ps[1])}
        r_tab = {(0,): row(pr[0]), (1,): row(pr[1])}
        w_tab = {(0, 0): row(pw[0]), (0, 1): row(pw[1]), (1, 0): row(pw[2]), (1, 1): row(pw[3])}
    else:
        c_tab = {(): {0: 0.5, 1: 0.5}}
        s_tab = {(0,): {0: 0.5, 1: 0.5}, (1,): {0: 0.9, 1: 0.1}}
        r_tab = {(0,): {0: 0.8, 1: 0.2}, (1,): {0: 0.2, 1: 0.8}}
        w_tab = {(0, 0): {0: 0.99, 1: 0.01}, (0, 1): {0: 0.1, 1: 0.9},
                 (1, 0): {0: 0.1, 1: 0.9}, (1, 1): {0: 0.01, 1: 0.99}}
    bn = BayesianNetwork()
    bn.add_node('C', cpt=c_tab)
    bn.add_node('S', cpt=s_tab, parent_names=['C'])
    bn.add_node('R', cpt=r_tab, parent_names=['C'])
    bn.add_node('W', cpt=w_tab, parent_names=['S', 'R'])
    return bn


def synthetic_network(n, seed=0, p_low=0.01, p_high=0.99):
    """n latent nodes Z0..Z{n-1} plus one observed leaf X.

    Z_k has parents [Z_{k-1}] (+ [Z_{k-2}] when k is even and k >= 2); X has parents
    [Z_{n-2}, Z_{n-1}] (just [Z_0] when n == 1).  Every P(node=1 | parents) ~ U(p_low, p_high)
    (default U(0.01, 0.99), the reference's ``random_p``, bayesian_network.py:323) from
    ``numpy.random.default_rng(seed)``, drawn node by node, parent configurations in
    lexicographic order.  Returns (bn, latent_names, observed_names, x_observation_dict).
    """
    rng = np.random.default_rng(seed)
    bn = BayesianNetwork()

    def table(num_parents):
        tab = {}
        for cfg in generate_all_binary_outcomes(num_parents):
            p1 = float(rng.uniform(p_low, p_high))
            tab[cfg] = {0: 1.0 - p1, 1: p1}
        return tab

    latents = [f"Z{k}" for k in range(n)]
    for k, name in enumerate(latents):
        pa = []
        if k >= 1:
            pa.append(latents[k - 1])
        if k >= 2 and k % 2 == 0:
            pa.append(latents[k - 2])
        bn.add_node(name, cpt=table(len(pa)), parent_names=pa or None)
    xpa = latents[-2:] if n >= 2 else latents[-1:]
    bn.add_node("X", cpt=table(len(xpa)), parent_names=xpa)
    return bn, latents, ["X"], {"X": 1}


ROLE_OBSERVED0, ROLE_OBSERVED1, ROLE_HIDDEN = -1, -2, -3


def pack_network(bn, latent_names, x_dict):
    """Flatten a (duck-typed) BayesianNetwork for ``bornvi_score_from_cpts``.

    Returns a dict of contiguous arrays:
      role      int32[V]   latent position (0..n-1, position 0 = MSB of the outcome index),
                           ROLE_OBSERVED0/1 for evidence, ROLE_HIDDEN for nodes summed out
      n_parents int32[V]
      parents   int32[V, MAX_PARENTS]   node indices, CPT key order
      cpt_off   int32[V]   offset (in doubles) of the node's table in ``cpt``
      cpt       float64[sum 2*2^|pa|]   table[config][value], config index = parent
                           values read as a binary number, first parent most significant;
                           both P(0|.) and P(1|.) are stored as given (never 1-p).
    Callable CPTs are tabulated by calling them on every parent configuration.
    """
    x_dict = dict(x_dict or {})
    V = len(bn.nodes)
    index = {nm: i for i, nm in enumerate(bn.nodes)}
    role = np.zeros(V, np.int32)
    n_par = np.zeros(V, np.int32)
    par = np.zeros((V, MAX_PARENTS), np.int32)
    off = np.zeros(V, np.int32)
    flat = []
    for i, nm in enumerate(bn.nodes):
        if nm in latent_names:
            role[i] = list(latent_names).index(nm)
        elif nm in x_dict:
            role[i] = ROLE_OBSERVED1 if int(x_dict[nm]) == 1 else ROLE_OBSERVED0
        else:
            role[i] = ROLE_HIDDEN
        pa = list(bn.parents[nm]) if nm in bn.parents else []
        if len(pa) > MAX_PARENTS:
            raise ValueError(f"node {nm} has {len(pa)} parents; at most {MAX_PARENTS} supported")
        n_par[i] = len(pa)
        for j, p in enumerate(pa):
            par[i, j] = index[p]
        off[i] = len(flat)
        entry = bn.cpts[nm]
        for cfg in generate_all_binary_outcomes(len(pa)):
            row = entry(cfg) if callable(entry) else entry.get(cfg)
            if row is None:
                raise ValueError(f"CPT entry for node {nm} with parent values {cfg} not found.")
            flat.extend([float(row[0]), float(row[1])])
    return {"role": role, "n_parents": n_par, "parents": par, "cpt_off": off,
            "cpt": np.asarray(flat, np.float64)}
