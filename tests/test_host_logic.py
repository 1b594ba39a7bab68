"""CPU-side checks: the C-ABI library loads and exports every symbol include/bornvi.h declares,
the circuit planner (C++ host code) is correct against the oracle via the NumPy plan emulator,
host helpers mirror the reference, and the product refuses to run without a GPU."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import circuit as oc, stein as os_
import plan_emulator as pe
from conftest import REPO, golden


def test_library_exports_every_declared_symbol():
    from tensornetworks_amd import _ext
    lib = _ext.lib()
    hdr = open(os.path.join(REPO, "include", "bornvi.h")).read()
    declared = set(re.findall(r"\b(bornvi_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in bornvi.h but not exported"
    assert declared == set(_ext.EXPORTED_SYMBOLS), declared ^ set(_ext.EXPORTED_SYMBOLS)
    assert lib.bornvi_version() == 100
    for ansatz, aid in _ext.ANSATZ_IDS.items():
        for n, L in [(3, 4), (8, 4), (16, 6), (20, 8)]:
            assert lib.bornvi_num_params(aid, n, L) == oc.num_params(ansatz, n, L)
            assert lib.bornvi_num_gates(aid, n, L) == len(oc.gate_list(ansatz, n, L))
    assert lib.bornvi_num_params(7, 3, 1) == -1 and lib.bornvi_num_gates(7, 3, 1) == -1


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_fails_loudly_without_gpu():
    from tensornetworks_amd import _ext, backend
    from tensornetworks_amd.quantum_born_machine import QuantumBornMachine
    with pytest.raises(_ext.BornviError):
        _ext.Handle(0)
    with pytest.raises(_ext.BornviError):
        backend.compute_device("cpu")
    bm = QuantumBornMachine(3, 1)
    with pytest.raises(_ext.BornviError, match="no CPU fallback"):
        bm.get_probabilities()


@pytest.mark.parametrize("ansatz", oc.ANSATZ_TYPES)
@pytest.mark.parametrize("n,L,kb", [(1, 1, 0), (2, 3, 0), (3, 4, 0), (5, 2, 0), (6, 3, 4), (7, 2, 5), (9, 2, 6),
                                    (10, 2, 7), (11, 1, 13), (12, 1, 8)])
def test_planner_against_oracle(ansatz, n, L, kb):
    """The pass / stage / micro-op program emitted by the C++ planner, interpreted on NumPy exactly
    as the HIP kernel interprets it, reproduces the oracle's q_theta."""
    from tensornetworks_amd import _ext
    W = _ext.plan_words(_ext.ANSATZ_IDS[ansatz], n, L, kb)
    rng = np.random.default_rng(n * 31 + L)
    th = rng.uniform(-np.pi, np.pi, oc.num_params(ansatz, n, L))
    q = pe.run_plan(W, pe.fused_matrices(W, th))
    np.testing.assert_allclose(q, oc.probs(ansatz, n, L, th), rtol=0, atol=1e-13)
    st = pe.plan_stats(W)
    assert st["gates"] == len(oc.gate_list(ansatz, n, L)) and st["k"] == min(n, kb or 13)


@pytest.mark.parametrize("ansatz", oc.ANSATZ_TYPES)
@pytest.mark.parametrize("n,L,kb", [(10, 1, 0), (11, 1, 0), (12, 3, 0), (13, 2, 11), (13, 1, 13), (14, 2, 12), (14, 3, 11)])
def test_fast_tables_against_oracle(ansatz, n, L, kb):
    """The per-(stage, tile, thread) tables the fast pass kernel reads (LDS slots with the CNOT index maps
    folded in, CZ sign bits, slot-offset bases, stage kinds), interpreted as that kernel interprets them."""
    from tensornetworks_amd import _ext
    aid = _ext.ANSATZ_IDS[ansatz]
    W = _ext.plan_words(aid, n, L, kb)
    F, offs = _ext.plan_fast_words(aid, n, L, kb)
    assert F is not None, "plan should be eligible for the fast kernel"
    th = np.random.default_rng(n * 7 + L).uniform(-np.pi, np.pi, oc.num_params(ansatz, n, L))
    q = pe.run_plan(W, pe.fused_matrices(W, th), fast=(F, offs))
    np.testing.assert_allclose(q, oc.probs(ansatz, n, L, th), rtol=0, atol=1e-13)


@pytest.mark.parametrize("ansatz", oc.ANSATZ_TYPES)
@pytest.mark.parametrize("n,L,kb,read_map", [(6, 3, 0, 0), (7, 2, 0, 0), (8, 4, 0, 0), (9, 3, 6, 0), (10, 2, 7, 0), (9, 1, 0, 0), (11, 2, 9, 0),
                                             (12, 3, 0, 1), (13, 2, 11, 0), (14, 2, 12, 1), (14, 3, 11, 1), (15, 3, 12, 0)])
def test_r3_plan_and_compact_tables_against_oracle(ansatz, n, L, kb, read_map):
    """The plan with 3 register wires per stage (8 amplitudes per thread, circuit_pass_r3_kernel): (i) its stage headers
    interpreted as the generic kernel does, (ii) its per-(tile row, thread) fast tables, (iii) its COMPACT tables -- every
    word = LANE[row][lane] ^ UNI[tile row][row][wave], sign rows with the bilinear lane x (wave, tile row) term -- run exactly
    as the kernel runs them (direct first / last stages on and off, support of |0..0> on and off) reproduce the oracle."""
    from tensornetworks_amd import _ext
    aid = _ext.ANSATZ_IDS[ansatz]
    flags = kb | _ext.R3 | (0x100 if read_map else 0)
    W = _ext.plan_words(aid, n, L, flags)
    assert int(W[pe.PH_R]) == 3 and int(W[pe.PH_THREADS]) == max(64, 1 << (int(W[pe.PH_K]) - 3))    # (tiles below 2^9: part of one wave)
    th = np.random.default_rng(n * 13 + L).uniform(-np.pi, np.pi, oc.num_params(ansatz, n, L))
    ref = oc.probs(ansatz, n, L, th)
    mats = pe.fused_matrices(W, th)
    np.testing.assert_allclose(pe.run_plan(W, mats), ref, rtol=0, atol=1e-13)
    F, offs = _ext.plan_fast_words(aid, n, L, flags)
    if F is not None:         # (the full tables keep the 16-amplitude kernel's rule "one matrix piece per thread"; the compact ones do not)
        np.testing.assert_allclose(pe.run_plan(W, mats, fast=(F, offs)), ref, rtol=0, atol=1e-13)
    Cw, coffs = _ext.plan_compact_words(aid, n, L, kb | (0x100 if read_map else 0))
    assert Cw is not None and len(coffs) == int(W[pe.PH_NPASSES])
    for direct, zs in ((3, True), (3, False), (0, False)):
        if read_map:
            continue          # (the compact emulator asserts stages without cross reads; those plans are covered on the GPU)
        np.testing.assert_allclose(pe.run_plan_compact(W, (Cw, coffs), mats, direct=direct, zero_support=zs), ref, rtol=0, atol=1e-13)


def test_r3_benchmark_sizes_are_eligible():
    """BASELINE configs 3 and 4 under the 8-amplitude kernel: 2^13 tiles (1024 threads), compact tables within the CU's LDS
    beside the tile; the planner's read map brings the stage count close to the bound of 3 gates per stage."""
    from tensornetworks_amd import _ext
    for n, L, gates in ((16, 6, 96), (20, 8, 160)):
        for rm, max_stages in ((0, 10 ** 6), (0x100, gates // 3 + 8)):
            W = _ext.plan_words(0, n, L, _ext.R3 | rm)
            st = pe.plan_stats(W)
            assert st["k"] == 13 and int(W[pe.PH_THREADS]) == 1024 and sum(st["stages"]) <= max_stages, st
        Cw, coffs = _ext.plan_compact_words(0, n, L, 0)
        assert Cw is not None and len(Cw) * 4 < 4 << 20


def test_zero_support_masks_leave_out_only_zeros():
    """Support of |0..0> (fast tables, FH_ZINFO): the INIT pass leaves out tiles nobody reads, the pass behind it does
    not load slots known to be zero.  The emulator writes NaN where a left-out tile would have gone and computes with 0
    where a slot is not loaded (asserting that only zeros or that poison lie there): the plan still reproduces the
    oracle -- and the masks are really in use at the sizes that matter."""
    from tensornetworks_amd import _ext
    used = 0
    for ansatz, n, L, kb in [("all_to_all", 13, 2, 10), ("hardware_efficient", 14, 3, 10), ("basic", 14, 3, 11),
                             ("hardware_efficient", 13, 3, 10), ("all_to_all", 14, 2, 11)]:
        aid = _ext.ANSATZ_IDS[ansatz]
        W = _ext.plan_words(aid, n, L, kb)
        F, offs = _ext.plan_fast_words(aid, n, L, kb)
        if F is None:                                     # (not eligible for the persistent kernel at this tile size)
            continue
        gmask, zslots = int(F[offs[0] + pe.FH_ZINFO]), int(F[offs[1] + pe.FH_ZINFO])
        assert (gmask == 0) == (zslots == 0)
        used += gmask != 0
        th = np.random.default_rng(n + L).uniform(-np.pi, np.pi, oc.num_params(ansatz, n, L))
        q = pe.run_plan(W, pe.fused_matrices(W, th), fast=(F, offs))
        np.testing.assert_allclose(q, oc.probs(ansatz, n, L, th), rtol=0, atol=1e-13)
    assert used >= 2
    for n, L in ((16, 6), (20, 8)):                      # the benchmark sizes: one resp. 8 of 2^(n-13) tiles written, 2 resp. 1 of 16 slots loaded
        F, offs = _ext.plan_fast_words(0, n, L, 0)
        gmask, zslots = int(F[offs[0] + pe.FH_ZINFO]), int(F[offs[1] + pe.FH_ZINFO])
        assert bin(gmask).count("1") == min(n - 13, 4) and bin(zslots).count("1") == 16 - (16 >> min(n - 13, 4))


def test_fast_tables_eligibility_and_kron():
    from tensornetworks_amd import _ext
    # the first / last stages of multi-pass plans go straight between HBM and registers (coalescing checked by
    # the table builder; the emulator checks the offsets against the ordinary tile fill / drain)
    Wd = _ext.plan_words(0, 14, 3, 11)
    Fd, od = _ext.plan_fast_words(0, 14, 3, 11)
    n_in = sum(int(Fd[int(o) + pe.FH_IN_TAB]) != 0 for o in od)
    n_out = sum(int(Fd[int(o) + pe.FH_OUT_TAB]) != 0 for o in od)
    assert n_in >= 1, (n_in, n_out, len(od))
    assert _ext.plan_fast_words(0, 8, 4, 0) == (None, None)          # tiles below 2^10: generic kernel
    assert _ext.plan_fast_words(0, 10, 2, 0) == (None, None)         # 6 stages x 16 matrix pieces > 64 threads
    assert _ext.plan_fast_words(0, 12, 2, 8) == (None, None)
    for n, L in [(16, 6), (20, 8)]:
        F, offs = _ext.plan_fast_words(0, n, L, 0)
        W = _ext.plan_words(0, n, L, 0)
        assert F is not None and len(offs) == int(W[3]) and len(F) * 4 < 64 << 20
    n, kb = 12, 10
    W = _ext.plan_words(-1, n, 0, kb)
    F, offs = _ext.plan_fast_words(-1, n, 0, kb)
    assert F is not None
    a = np.exp(-1.0 / n)
    M = np.array([[1, a], [a, 1]], dtype=np.complex128)
    rng = np.random.default_rng(n)
    v = rng.normal(size=2 ** n) + 1j * rng.normal(size=2 ** n)
    out = pe.run_plan(W, [M], state_in=v, fast=(F, offs))
    ref = os_.kbase_apply(v.real, n, a) + 1j * os_.kbase_apply(v.imag, n, a)
    np.testing.assert_allclose(out, ref, rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("n,kb", [(1, 0), (3, 0), (6, 4), (9, 5), (10, 7), (12, 13)])
def test_kron_plan_against_oracle(n, kb):
    """State-in / state-out program for K_base = M^{(x) n} (matrix-free Stein mat-vec)."""
    from tensornetworks_amd import _ext
    W = _ext.plan_words(-1, n, 0, kb)
    a = np.exp(-1.0 / n)
    M = np.array([[1, a], [a, 1]], dtype=np.complex128)
    rng = np.random.default_rng(n)
    v = rng.normal(size=2 ** n) + 1j * rng.normal(size=2 ** n)
    out = pe.run_plan(W, [M], state_in=v)
    ref = os_.kbase_apply(v.real, n, a) + 1j * os_.kbase_apply(v.imag, n, a)
    np.testing.assert_allclose(out, ref, rtol=1e-13, atol=1e-13)


def test_full_size_plans_are_well_formed():
    from tensornetworks_amd import _ext
    for ansatz_id in (0, 1, 2):
        for n, L in [(16, 6), (20, 8)]:
            W = _ext.plan_words(ansatz_id, n, L, 0)
            st = pe.plan_stats(W)
            # tile size of a multi-tile state: 2^13 wherever the fast kernel can run it (make_plan)
            # (all_to_all at n = 20 carries CZ sign tables in every stage: tile + table rows exceed 160 KiB of LDS -> 2^11)
            assert st["k"] == (11 if (ansatz_id, n) == (1, 20) else 13)
            assert _ext.plan_fast_words(ansatz_id, n, L, 0)[0] is not None
            p11 = pe.plan_stats(_ext.plan_words(ansatz_id, n, L, 11))["passes"]
            p13 = pe.plan_stats(_ext.plan_words(ansatz_id, n, L, 13))["passes"]
            assert st["passes"] == (p13 if st["k"] == 13 else p11) and p13 <= p11
            assert 2 <= st["passes"] <= 3 * L + 2
            assert int(W[8]) == len(W)
    assert _ext.lib().bornvi_plan_describe(0, 31, 1, 0, None, 0) == -1       # n out of range
    assert _ext.lib().bornvi_plan_describe(9, 4, 1, 0, None, 0) == -1        # unknown ansatz


def test_utils_and_network_helpers():
    from tensornetworks_amd import utils as u
    from tensornetworks_amd.bayesian_network import (BayesianNetwork, get_sprinkler_network, pack_network,
                                                      synthetic_network, joint_table)
    assert u.generate_all_binary_outcomes(0) == [()]
    assert u.generate_all_binary_outcomes(1) == [(0,), (1,)]
    assert u.generate_all_binary_outcomes(3) == os_.generate_all_binary_outcomes(3)
    assert all(u.outcome_index(z) == i for i, z in enumerate(u.generate_all_binary_outcomes(5)))
    p1 = {'00': 0.25, '01': 0.25, '10': 0.25, '11': 0.25}
    p2 = {'00': 0.5, '01': 0.1, '10': 0.1, '11': 0.3}
    assert abs(u.calculate_tvd(p1, p2) - 0.3) < 1e-15                                  # utils.py:96-102
    assert abs(u.calculate_tvd(np.array([.25] * 4), np.array([.5, .1, .1, .3])) - 0.3) < 1e-15
    with pytest.raises(ValueError):
        u.calculate_tvd(np.zeros(3), np.zeros(4))
    with pytest.raises(TypeError):
        u.calculate_tvd(p1, np.zeros(4))
    bn = get_sprinkler_network(False)
    assert bn.nodes == ['C', 'S', 'R', 'W'] and bn.parents['W'] == ['S', 'R']
    assert abs(bn.get_joint_probability((1, 0, 1, 1)) - 0.5 * 0.9 * 0.8 * 0.9) < 1e-16
    with pytest.raises(ValueError):
        bn.add_node('C', cpt={})
    with pytest.raises(ValueError):
        BayesianNetwork().add_node('X', cpt={}, parent_names=['nope'])
    with pytest.raises(ValueError):
        bn.get_joint_probability((1, 0))
    g = golden("sprinkler_w1.npz")
    np.testing.assert_array_equal(joint_table(bn, ['C', 'S', 'R'], {'W': 1}), g["pxz"])
    pk = pack_network(bn, ['C', 'S', 'R'], {'W': 1})
    for k in pk:
        np.testing.assert_array_equal(pk[k], g["pack_" + k])
    sb, lat, obs, x = synthetic_network(6, 0)
    assert lat == [f"Z{i}" for i in range(6)] and obs == ["X"] and x == {"X": 1}
    assert sb.parents["Z4"] == ["Z3", "Z2"] and sb.parents["Z5"] == ["Z4"] and sb.parents["X"] == ["Z4", "Z5"]
    sb2, *_ = synthetic_network(6, 0)
    assert sb.cpts == sb2.cpts                                                          # seeded
    hidden = pack_network(bn, ['C', 'R'], {'W': 0})
    assert hidden["role"].tolist() == [0, -3, 1, -1]


def test_shard_ranges_cover_all_parameters():
    from tensornetworks_amd.paramshift_shard import shard_range
    for P in (0, 1, 7, 36, 96, 288, 480):
        for W in (1, 2, 3, 4, 8):
            got = []
            for r in range(W):
                lo, hi = shard_range(P, r, W)
                assert 0 <= lo <= hi <= P
                got += list(range(lo, hi))
            assert got == list(range(P))
    assert shard_range(288, 3, 8) == (108, 144) and shard_range(480, 7, 8) == (420, 480)


def test_interleaved_parameter_deal_and_gather_order():
    """shard_params deals the parameters rank, rank + W, ...; all_gather_grad's re-ordering (rank-major gathered
    buffer -> parameter order) is its exact inverse."""
    import torch
    from tensornetworks_amd.paramshift_shard import shard_params
    for P in (0, 1, 7, 36, 96, 288, 480):
        for W in (1, 2, 3, 4, 8):
            chunk = -(-P // W) if P else 0
            owned = [list(range(*shard_params(P, r, W))) for r in range(W)]
            assert sorted(sum(owned, [])) == list(range(P))
            assert all(len(o) <= chunk for o in owned) and max(map(len, owned)) - min(map(len, owned)) <= 1
            if P == 0:
                continue
            gathered = torch.full((W * chunk,), -1.0, dtype=torch.float64)       # what all_gather_into_tensor delivers
            for r in range(W):
                gathered[r * chunk: r * chunk + len(owned[r])] = torch.tensor(owned[r], dtype=torch.float64)
            full = gathered.view(W, chunk).t().reshape(-1)[:P]
            assert full.tolist() == [float(p) for p in range(P)]



def test_planner_property_random_configurations():
    """Property test: for random (ansatz, n, layers, tile size) the emitted plan reproduces the oracle."""
    from hypothesis import given, settings, strategies as st
    from tensornetworks_amd import _ext

    @settings(max_examples=30, deadline=None, derandomize=True)
    @given(st.sampled_from(oc.ANSATZ_TYPES), st.integers(1, 9), st.integers(0, 3), st.integers(4, 13), st.integers(0, 10 ** 6),
           st.booleans())
    def check(ansatz, n, L, kb, seed, read_map):
        # (bit 8 of the tile-size argument: the planner's `read_map` option -- phase-0 CNOTs on thread-held wires, stages
        # flagged STAGE_CROSS_READ; off in the product, kept correct)
        W = _ext.plan_words(_ext.ANSATZ_IDS[ansatz], n, L, kb | (0x100 if read_map else 0))
        th = np.random.default_rng(seed).uniform(-np.pi, np.pi, oc.num_params(ansatz, n, L))
        q = pe.run_plan(W, pe.fused_matrices(W, th))
        np.testing.assert_allclose(q, oc.probs(ansatz, n, L, th), rtol=0, atol=1e-13)
        assert abs(q.sum() - 1) < 1e-13
    check()


def test_c_port_matches_numpy_oracle():
    from oracle import cpu_port as cp
    if not cp.available():
        pytest.skip("oracle/_build/libcpu_port.so not built (run __graft_entry__.build())")
    rng = np.random.default_rng(1)
    for ansatz in oc.ANSATZ_TYPES:
        th = rng.uniform(-3, 3, (2, oc.num_params(ansatz, 7, 2)))
        q = cp.circuit_probs(ansatz, 7, 2, th)
        for b in range(2):
            np.testing.assert_allclose(q[b], oc.probs(ansatz, 7, 2, th[b]), rtol=0, atol=1e-15)
    g = golden("synthetic_n6_s0.npz")
    K = cp.gram_rows(g["S"], 6, 1.0, 0, 64)
    np.testing.assert_allclose(K, g["K"], rtol=0, atol=3e-15 * np.abs(g["K"]).max())
    q = rng.random(64); q /= q.sum()
    y, part = cp.gemv_rows(K[8:24], 6, 8, 24, q)
    np.testing.assert_allclose(y, K[8:24] @ q, rtol=1e-14)
    assert abs(part - q[8:24] @ (K[8:24] @ q)) < 1e-12 * abs(part)
    pr, used = cp.paramshift_probs("basic", 5, 2, th[0][:20], 2, 4, True)
    tp = th[0][:20].copy(); tp[3] -= np.pi / 2
    np.testing.assert_allclose(pr[4], oc.probs("basic", 5, 2, tp), atol=1e-15)
    assert pr.shape == (5, 32) and used >= 1
    # matrix-free K_p q (the CPU baseline's contraction at n = 20) == dense closed form == NumPy Kronecker oracle
    from oracle import stein as os_
    yk, k2 = cp.kron_matvec(g["S"], q, 6, 1.0)
    np.testing.assert_allclose(yk, g["K"] @ q, rtol=0, atol=1e-13 * np.abs(g["K"] @ q).max())
    np.testing.assert_allclose(yk, os_.stein_matvec_kron(g["S"], q, 6, 1.0), rtol=0, atol=1e-13 * np.abs(yk).max())
    assert abs(k2 - q @ g["K"] @ q) <= 1e-12 * abs(k2)


def test_param_first_pass_from_plan_words():
    """Every parameter is touched by some pass, single-pass plans share nothing, and with several passes the early
    layers' parameters come first (parameters are numbered layer by layer): the ordering prefix sharing relies on."""
    from tensornetworks_amd import _ext
    W = _ext.plan_words(_ext.ANSATZ_IDS["hardware_efficient"], 10, 3)
    assert int(W[3]) == 1 and not _ext.plan_param_first_pass(W).any()
    for ansatz, n, L in (("hardware_efficient", 16, 6), ("basic", 15, 3), ("all_to_all", 14, 2)):
        W = _ext.plan_words(_ext.ANSATZ_IDS[ansatz], n, L)
        fp = _ext.plan_param_first_pass(W)
        npass = int(W[3])
        assert fp.shape == (int(W[5]),) and fp.min() == 0 and fp.max() <= npass - 1
        assert len(set(fp.tolist())) > 1
        per_layer = fp.reshape(L, -1) if ansatz != "all_to_all" else None
        if per_layer is not None:       # the mean first pass grows with the layer
            m = per_layer.mean(axis=1)
            assert all(m[i] <= m[i + 1] for i in range(L - 1))


def test_first_pass_rule_is_the_librarys():
    """The parser bench.py uses for its byte accounting (_ext.plan_param_first_pass on the serialised plan) and the
    planner's own Plan::param_first_pass (what prefix sharing orders the batch by) are the same numbers."""
    import ctypes as C
    from tensornetworks_amd import _ext
    L_ = _ext.lib()
    for ansatz, n, L in (("hardware_efficient", 16, 6), ("hardware_efficient", 20, 8), ("basic", 15, 3), ("all_to_all", 14, 2),
                         ("hardware_efficient", 10, 3)):
        aid = _ext.ANSATZ_IDS[ansatz]
        P = L_.bornvi_plan_param_first_pass(aid, n, L, 0, None, 0)
        assert P == len(_ext.plan_param_first_pass(_ext.plan_words(aid, n, L)))
        buf = (C.c_int * P)()
        assert L_.bornvi_plan_param_first_pass(aid, n, L, 0, buf, P) == P
        assert list(buf) == _ext.plan_param_first_pass(_ext.plan_words(aid, n, L)).tolist()
    assert L_.bornvi_plan_param_first_pass(99, 8, 2, 0, None, 0) == -1


def test_trainer_rejects_mismatched_latent_count():
    """qbm_num_latent_vars != len(latent_vars_names) would make the device kernels read S [2^m, m] as [2^n, n]
    (an out-of-bounds read): refused in the constructor, before anything touches a device."""
    from tensornetworks_amd.bayesian_network import get_sprinkler_network
    from tensornetworks_amd.ksd_vi_quantum import KSDVariationalInference
    with pytest.raises(ValueError, match="must equal len"):
        KSDVariationalInference(get_sprinkler_network(False), ['C', 'S', 'R'], ['W'], qbm_num_latent_vars=4)


def test_bench_refuses_a_world_size_that_is_not_gpus(monkeypatch):
    """bench.py --gpus N under a launcher that started a different number of ranks exits non-zero (before any GPU call)
    instead of measuring the wrong job with a note on stderr."""
    import bench
    monkeypatch.setenv("WORLD_SIZE", "3")
    monkeypatch.setenv("RANK", "0")
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "2"])
    assert "WORLD_SIZE=3" in str(e.value)


def test_reinforce_step_against_the_reference_trace():
    """The Born-machine (REINFORCE) step of the adversarial trainer against a trace captured from the reference's own
    train() (adversarial_vi.py:184-231; tests/golden/make_golden.py: reinforce_trace -- sample indices, classifier logits,
    log p(x|z), log q, running baseline and loss_q of 4 epochs): the build's reward / baseline / loss arithmetic, fed the
    reference's per-step inputs, reproduces the reference's baseline and loss_q; log q comes from the build's own
    expression log(clamp(q, 1e-9))[index] on the reference's q (quantum_born_machine.py:180-201)."""
    import torch
    from tensornetworks_amd.adversarial_vi import AdversarialVariationalInference as A
    g = golden("reinforce_trace.npz")
    decay = float(g["baseline_decay"])
    baseline = torch.zeros(())
    for e in range(len(g["loss_q"])):
        logits, log_p = torch.tensor(g["logits"][e]), torch.tensor(g["log_p"][e])
        idx = torch.tensor(g["idx"][e], dtype=torch.long)
        log_q = torch.log(torch.tensor(g["q"][e]).clamp(min=1e-9))[idx]
        np.testing.assert_allclose(log_q.numpy(), g["log_q"][e], rtol=1e-6, atol=1e-7)
        rr = A._reinforce_reward(logits, log_p, baseline, e == 0, decay)
        loss_q = A._reinforce_loss(log_q, rr)
        assert abs(baseline.item() - g["baseline"][e]) <= 1e-6 * abs(g["baseline"][e])
        assert abs(loss_q.item() - g["loss_q"][e]) <= 1e-6 * max(1.0, abs(g["loss_q"][e])), (e, loss_q.item(), g["loss_q"][e])


def test_cosine_schedule_table_follows_torch():
    """DeviceAdam tabulates CosineAnnealingLR by its closed form: equal to the scheduler's own (recursive) values to
    rounding, beyond T_max as well (the bench replays more steps than T_max)."""
    import torch
    from tensornetworks_amd.ksd_vi_quantum import cosine_annealing_lr
    for lr0, T, eta in ((0.05, 7, 0.005), (0.005, 1285, 0.0005), (0.1, 1, 0.0)):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([p], lr=lr0)
        sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=T, eta_min=eta)
        want = [lr0]
        for _ in range(min(3 * T + 5, 4000)):
            opt.step()
            sch.step()
            want.append(sch.get_last_lr()[0])
        got = cosine_annealing_lr(np.arange(len(want)), lr0, T, eta)
        np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-15)
        assert cosine_annealing_lr(0, lr0, T, eta) == lr0
