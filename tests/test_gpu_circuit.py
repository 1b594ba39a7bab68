"""GPU parity of the circuit engine (through the C ABI) against the CPU oracle.

Tolerance: q_theta within 1e-10 relative / 1e-14 absolute of the oracle (north_star asks 1e-6
relative, fp64); fp64 throughout."""
import math

import numpy as np
import pytest
import torch

from oracle import circuit as oc

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-10, 1e-14


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda", 0)


@pytest.fixture(params=[3, 4], ids=["r3", "r4"])
def be(dev, request):
    """Every test of this file runs under both persistent pass kernels: 8 amplitudes per thread (reg_wires = 3,
    circuit_pass_r3_kernel, the default) and 16 per thread (reg_wires = 4, circuit_pass_fast_kernel, also the fallback of
    plans the first cannot run)."""
    from tensornetworks_amd import backend
    default_r = backend.get_option(dev, "reg_wires")
    backend.set_option(dev, "reg_wires", request.param)
    yield backend
    backend.set_option(dev, "tile_bits", 13)          # restore the planner defaults
    backend.set_option(dev, "tile_bits_multi", 0)     # (0 = automatic: 2^13 tiles where the persistent kernel can run them)
    backend.set_option(dev, "read_map", -1)           # (-1 = by the kernel: on with 8 amplitudes per thread)
    backend.set_option(dev, "reg_wires", default_r)


def gpu_probs(be, dev, ansatz, n, L, thetas):
    t = torch.as_tensor(np.atleast_2d(thetas), dtype=torch.float64, device=dev).contiguous()
    return be.circuit_probs(ansatz, n, L, t).cpu().numpy()


@pytest.mark.parametrize("ansatz", oc.ANSATZ_TYPES)
@pytest.mark.parametrize("n,L", [(1, 1), (2, 2), (3, 4), (4, 1), (5, 3), (8, 4), (10, 2), (12, 4), (13, 2)])
def test_probs_match_oracle_single_pass(be, dev, ansatz, n, L):
    rng = np.random.default_rng(100 * n + L)
    th = rng.uniform(-np.pi, np.pi, (3, oc.num_params(ansatz, n, L)))
    q = gpu_probs(be, dev, ansatz, n, L, th)
    for b in range(3):
        np.testing.assert_allclose(q[b], oc.probs(ansatz, n, L, th[b]), rtol=RTOL, atol=ATOL)
        assert abs(q[b].sum() - 1.0) < 1e-13


@pytest.mark.parametrize("ansatz", oc.ANSATZ_TYPES)
@pytest.mark.parametrize("n,L,kb", [(6, 3, 4), (9, 3, 6), (10, 2, 7), (12, 2, 9), (14, 3, 13), (15, 2, 12), (16, 2, 13)])
def test_probs_match_oracle_multi_pass(be, dev, ansatz, n, L, kb):
    """Tiles smaller than the state: several passes with HBM re-layout in between."""
    be.set_option(dev, "tile_bits", kb)
    rng = np.random.default_rng(7 * n + L + kb)
    th = rng.uniform(-np.pi, np.pi, (2, oc.num_params(ansatz, n, L)))
    q = gpu_probs(be, dev, ansatz, n, L, th)
    for b in range(2):
        np.testing.assert_allclose(q[b], oc.probs(ansatz, n, L, th[b]), rtol=RTOL, atol=ATOL)


def test_known_answers(be, dev):
    # theta = 0 -> uniform (hardware_efficient / all_to_all), delta (basic)
    for ansatz in ("hardware_efficient", "all_to_all"):
        q = gpu_probs(be, dev, ansatz, 6, 3, np.zeros(oc.num_params(ansatz, 6, 3)))[0]
        np.testing.assert_allclose(q, np.full(64, 1 / 64), atol=1e-15)
    q = gpu_probs(be, dev, "basic", 5, 2, np.zeros(oc.num_params("basic", 5, 2)))[0]
    e = np.zeros(32); e[0] = 1
    np.testing.assert_allclose(q, e, atol=1e-15)
    # n = 1 closed form, independent of the RX / RZ angles
    q = gpu_probs(be, dev, "hardware_efficient", 1, 1, [0.3, 0.9, -1.1])[0]
    np.testing.assert_allclose(q, [(1 - math.sin(0.9)) / 2, (1 + math.sin(0.9)) / 2], atol=1e-15)
    # n = 2 basic: pins CNOT direction and MSB ordering
    t = [0.4, 0.2, 1.3, 0.5]
    q = gpu_probs(be, dev, "basic", 2, 1, t)[0]
    c0, s0 = math.cos(t[0] / 2) ** 2, math.sin(t[0] / 2) ** 2
    c1, s1 = math.cos(t[2] / 2) ** 2, math.sin(t[2] / 2) ** 2
    np.testing.assert_allclose(q, [c0 * c1, c0 * s1, s0 * s1, s0 * c1], atol=1e-15)


def test_zero_layers_and_large_batch(be, dev):
    q = gpu_probs(be, dev, "hardware_efficient", 4, 0, np.zeros((2, 0)))
    np.testing.assert_allclose(q, np.full((2, 16), 1 / 16), atol=1e-15)      # only the Hadamards
    q = gpu_probs(be, dev, "basic", 5, 0, np.zeros((3, 0)))                  # no gate at all: |0..0> (found by
    e0 = np.zeros(32); e0[0] = 1.0                                           # tests/test_gpu_properties.py: this divided by 0)
    np.testing.assert_array_equal(q, np.tile(e0, (3, 1)))
    # an empty batch is not an error (the buffers of an empty tensor are null pointers)
    P = oc.num_params("all_to_all", 6, 2)
    assert tuple(be.circuit_probs("all_to_all", 6, 2, torch.zeros((0, P), dtype=torch.float64, device=dev)).shape) == (0, 64)
    rng = np.random.default_rng(1)
    th = rng.uniform(-1, 1, (300, oc.num_params("basic", 7, 2)))
    q = gpu_probs(be, dev, "basic", 7, 2, th)
    np.testing.assert_allclose(q, oc.probs_batched("basic", 7, 2, th), rtol=RTOL, atol=ATOL)


def test_workspace_chunking(be, dev, monkeypatch):
    """A workspace that holds fewer circuits than the batch makes the library loop over chunks."""
    n, L = 14, 1
    be.set_option(dev, "tile_bits", 12)            # multi-pass => states live in the workspace
    rng = np.random.default_rng(3)
    th = rng.uniform(-1, 1, (5, oc.num_params("hardware_efficient", n, L)))
    full = gpu_probs(be, dev, "hardware_efficient", n, L, th)
    monkeypatch.setattr(be, "WORKSPACE_CAP", 2 * (2 * 16 * (1 << n)) + 40000)   # room for ~2 circuits
    be.release_workspaces()
    chunked = gpu_probs(be, dev, "hardware_efficient", n, L, th)
    np.testing.assert_array_equal(full, chunked)
    be.release_workspaces()


@pytest.mark.parametrize("ansatz", oc.ANSATZ_TYPES)
@pytest.mark.parametrize("n,L", [(3, 2), (5, 2), (8, 1)])
def test_paramshift_grad_matches_oracle(be, dev, ansatz, n, L):
    rng = np.random.default_rng(n + L)
    P = oc.num_params(ansatz, n, L)
    th = rng.uniform(-1, 1, P)
    w = rng.normal(size=2 ** n)
    g_o = oc.paramshift_vjp(ansatz, n, L, th, w)
    tht = torch.as_tensor(th, device=dev)
    wt = torch.as_tensor(w, device=dev)
    g = be.paramshift_grad(ansatz, n, L, tht, wt, 0, P).cpu().numpy()
    np.testing.assert_allclose(g, g_o, rtol=1e-9, atol=1e-12 * np.abs(g_o).max())
    # a slice of the parameters (what one rank of a sharded step computes)
    lo, hi = P // 3, 2 * P // 3
    gs = be.paramshift_grad(ansatz, n, L, tht, wt, lo, hi).cpu().numpy()
    np.testing.assert_array_equal(gs, g[lo:hi])
    # shifted probabilities layout: base, then (+p, -p)
    pr = be.paramshift_probs(ansatz, n, L, tht, 1, 3, include_base=True).cpu().numpy()
    np.testing.assert_allclose(pr[0], oc.probs(ansatz, n, L, th), rtol=RTOL, atol=ATOL)
    tp = th.copy(); tp[2] += np.pi / 2
    tm = th.copy(); tm[2] -= np.pi / 2
    np.testing.assert_allclose(pr[3], oc.probs(ansatz, n, L, tp), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(pr[4], oc.probs(ansatz, n, L, tm), rtol=RTOL, atol=ATOL)


def test_unfused_gate_kernels(be, dev):
    n, B = 9, 5
    rng = np.random.default_rng(0)
    psi = rng.normal(size=(B, 2 ** n)) + 1j * rng.normal(size=(B, 2 ** n))
    psi /= np.linalg.norm(psi, axis=1, keepdims=True)
    for wire in (0, 4, 8):
        U = oc.matrix_1q("RY", 0.3) @ oc.matrix_1q("RZ", -0.7) @ oc.matrix_1q("RX", 1.1)
        st = torch.as_tensor(psi, device=dev).contiguous()
        be.gate1q_apply(st, n, wire, U)
        ref = np.stack([oc.apply_1q(p.reshape((2,) * n), U, wire).reshape(-1) for p in psi])
        np.testing.assert_allclose(st.cpu().numpy(), ref, rtol=0, atol=1e-15)
    for c, t in ((0, 8), (8, 0), (3, 4)):
        st = torch.as_tensor(psi, device=dev).contiguous()
        be.cnot_apply(st, n, c, t)
        ref = np.stack([oc.apply_cnot(p.reshape((2,) * n), c, t).reshape(-1) for p in psi])
        np.testing.assert_array_equal(st.cpu().numpy(), ref)
    st = torch.as_tensor(psi, device=dev).contiguous()
    np.testing.assert_allclose(be.born_probs(st, n).cpu().numpy(), np.abs(psi) ** 2, rtol=1e-15)


def test_full_size_properties(be, dev):
    """BASELINE config 3 size (n = 16, L = 6): normalisation, agreement of the fused engine with a
    gate-by-gate run of the un-fused kernels, and parameter-shift == finite difference."""
    n, L, ansatz = 16, 6, "hardware_efficient"
    P = oc.num_params(ansatz, n, L)
    g = torch.Generator().manual_seed(0)
    th = (0.1 * torch.randn(P, generator=g, dtype=torch.float32)).double()
    tht = th.to(dev)
    q = be.circuit_probs(ansatz, n, L, tht.view(1, -1))[0]
    assert abs(q.sum().item() - 1.0) < 1e-12
    # gate-by-gate on the device with the un-fused kernels (independent code path)
    st = torch.zeros((1, 2 ** n), dtype=torch.complex128, device=dev)
    st[0, 0] = 1.0
    thn = th.numpy()
    for kind, wires, p in oc.gate_list(ansatz, n, L):
        if kind == "CNOT":
            be.cnot_apply(st, n, wires[0], wires[1])
        elif kind == "CZ":                                       # CZ = H_t CNOT H_t
            be.gate1q_apply(st, n, wires[1], oc.matrix_1q("H"))
            be.cnot_apply(st, n, wires[0], wires[1])
            be.gate1q_apply(st, n, wires[1], oc.matrix_1q("H"))
        else:
            be.gate1q_apply(st, n, wires[0], oc.matrix_1q(kind, None if p is None else thn[p]))
    np.testing.assert_allclose(q.cpu().numpy(), be.born_probs(st, n)[0].cpu().numpy(), rtol=1e-9, atol=1e-16)
    np.testing.assert_allclose(q.cpu().numpy(), oc.probs(ansatz, n, L, thn), rtol=1e-9, atol=1e-16)
    # parameter-shift gradient of a linear functional == central finite difference (3 parameters)
    w = torch.randn(2 ** n, generator=g, dtype=torch.float64).to(dev)
    for p in (0, P // 2, P - 1):
        gp = be.paramshift_grad(ansatz, n, L, tht, w, p, p + 1).item()
        h = 1e-5
        tp = tht.clone(); tp[p] += h
        tm = tht.clone(); tm[p] -= h
        qq = be.circuit_probs(ansatz, n, L, torch.stack([tp, tm]))
        fd = ((qq[0] - qq[1]) @ w).item() / (2 * h)
        assert abs(gp - fd) < 1e-7 * max(1.0, abs(fd))


@pytest.mark.parametrize("ansatz", oc.ANSATZ_TYPES)
def test_full_depth_n16_against_c_port(be, dev, ansatz):
    """All three ansaetze at n = 16, L = 6 (multi-tile plans, default tile size) against the oracle's C port."""
    from oracle import cpu_port as cp
    if not cp.available():
        pytest.skip("oracle/_build/libcpu_port.so not built")
    n, L = 16, 6
    rng = np.random.default_rng(42)
    th = rng.uniform(-np.pi, np.pi, (3, oc.num_params(ansatz, n, L)))
    q = gpu_probs(be, dev, ansatz, n, L, th)
    np.testing.assert_allclose(q, cp.circuit_probs(ansatz, n, L, th), rtol=1e-9, atol=1e-16)


def test_n20_circuit_and_kron_matvec(be, dev):
    """BASELINE config 4 size (n = 20): circuits (L = 1, 256 tiles per state) and the matrix-free Stein
    mat-vec against the CPU oracle."""
    from oracle import stein as os_
    from tensornetworks_amd.bayesian_network import synthetic_network, pack_network
    n, L = 20, 1
    rng = np.random.default_rng(20)
    th = rng.uniform(-np.pi, np.pi, (2, oc.num_params("hardware_efficient", n, L)))
    q = gpu_probs(be, dev, "hardware_efficient", n, L, th)
    for b in range(2):
        np.testing.assert_allclose(q[b], oc.probs("hardware_efficient", n, L, th[b]), rtol=1e-9, atol=1e-17)
    bn, lat, obs, x = synthetic_network(n, 0)
    S, pxz = be.score_from_packed(pack_network(bn, lat, x), n, dev)
    qq = torch.as_tensor(q[0], device=dev)
    k2, y = be.stein_matvec_kron(S, qq, n, 1.0)
    y_o = os_.stein_matvec_kron(S.cpu().numpy(), q[0], n)
    scale = np.abs(y_o).max()
    np.testing.assert_allclose(y.cpu().numpy(), y_o, rtol=0, atol=1e-10 * scale)
    assert abs(k2.item() - float(q[0] @ y_o)) <= 1e-9 * abs(float(q[0] @ y_o))


@pytest.mark.parametrize("ansatz", oc.ANSATZ_TYPES)
def test_persistent_tile_loop(be, dev, ansatz):
    """More tiles than co-resident workgroups: every workgroup of the fast pass kernel walks over several tiles
    (next-tile prefetch in flight, hand-counted vmcnt waits, stage tables kept in LDS).  All circuits of the batch
    use one of two parameter vectors, so every output row must equal the oracle's row for its parameters."""
    n, L, kb = 14, 2, 11
    be.set_option(dev, "tile_bits", kb)
    be.set_option(dev, "fast_workgroups_per_cu", 1)        # 256 workgroups for 100 x 8 tiles
    try:
        rng = np.random.default_rng(5)
        th2 = rng.uniform(-np.pi, np.pi, (2, oc.num_params(ansatz, n, L)))
        pick = rng.integers(0, 2, 100)
        q = gpu_probs(be, dev, ansatz, n, L, th2[pick])
        ref = [oc.probs(ansatz, n, L, th2[0]), oc.probs(ansatz, n, L, th2[1])]
        for b in range(100):
            np.testing.assert_allclose(q[b], ref[pick[b]], rtol=RTOL, atol=ATOL)
    finally:
        be.set_option(dev, "fast_workgroups_per_cu", 0)


@pytest.mark.parametrize("ansatz", oc.ANSATZ_TYPES)
@pytest.mark.parametrize("n,L,kb", [(14, 3, 11), (15, 2, 11), (16, 2, 12)])
def test_prefix_sharing_is_bit_identical(be, dev, ansatz, n, L, kb, monkeypatch):
    """A shifted circuit of a parameter-shift batch starts from the base circuit's state before the first pass its
    parameter touches (option prefix_share, opt-in).  Same passes, same matrices, same order of operations => the rows
    are BITWISE those of the batch that runs every circuit from |0..0>; and the rows agree with the oracle."""
    be.set_option(dev, "tile_bits", kb)
    P = oc.num_params(ansatz, n, L)
    rng = np.random.default_rng(11 * n + L)
    th = rng.uniform(-np.pi, np.pi, P)
    tht = torch.as_tensor(th, device=dev)
    lo, hi = P // 5, P - P // 7
    try:
        be.set_option(dev, "prefix_share", 0)
        ref_full = be.paramshift_probs(ansatz, n, L, tht, 0, P, include_base=True).cpu().numpy()
        ref_part = be.paramshift_probs(ansatz, n, L, tht, lo, hi, include_base=False).cpu().numpy()
        be.set_option(dev, "prefix_share", 1)
        got_full = be.paramshift_probs(ansatz, n, L, tht, 0, P, include_base=True).cpu().numpy()
        got_part = be.paramshift_probs(ansatz, n, L, tht, lo, hi, include_base=False).cpu().numpy()
        np.testing.assert_array_equal(got_full, ref_full)
        np.testing.assert_array_equal(got_part, ref_part)
        # a workspace for ~5 circuits: several chunks, each led by its own copy of the base circuit
        monkeypatch.setattr(be, "WORKSPACE_CAP", 5 * (2 * 16 * (1 << n)) + 200000)
        be.release_workspaces()
        got_chunked = be.paramshift_probs(ansatz, n, L, tht, 0, P, include_base=True).cpu().numpy()
        np.testing.assert_array_equal(got_chunked, ref_full)
        got_chunked = be.paramshift_probs(ansatz, n, L, tht, lo, hi, include_base=False).cpu().numpy()
        np.testing.assert_array_equal(got_chunked, ref_part)
    finally:
        be.set_option(dev, "prefix_share", 0)
        be.release_workspaces()
    # oracle on a few rows: base, first / last parameter, both signs
    np.testing.assert_allclose(got_full[0], oc.probs(ansatz, n, L, th), rtol=RTOL, atol=ATOL)
    for p_ in (0, P // 2, P - 1):
        for sgn, row in ((+1, 1 + 2 * p_), (-1, 2 + 2 * p_)):
            t2 = th.copy(); t2[p_] += sgn * np.pi / 2
            np.testing.assert_allclose(got_full[row], oc.probs(ansatz, n, L, t2), rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("ansatz", oc.ANSATZ_TYPES)
@pytest.mark.parametrize("n,L,kb,share", [(14, 3, 11, 0), (15, 2, 13, 0), (14, 2, 12, 1)])
def test_alternate_walk_is_bit_identical(be, dev, ansatz, n, L, kb, share):
    """Option alternate_walk (default on): odd passes walk the tiles of the batch from the last to the first (a pass
    starts on the states the previous one wrote last).  Only the order of independent tiles changes: rows BITWISE equal
    to the forward walk, with and without prefix sharing, for several tiles per workgroup (a batch of 2P + 1 circuits);
    the forward rows agree with the oracle."""
    be.set_option(dev, "tile_bits", kb)
    P = oc.num_params(ansatz, n, L)
    th = np.random.default_rng(5 * n + L).uniform(-np.pi, np.pi, P)
    tht = torch.as_tensor(th, device=dev)
    try:
        be.set_option(dev, "prefix_share", share)
        be.set_option(dev, "alternate_walk", 0)
        fwd = be.paramshift_probs(ansatz, n, L, tht, 0, P, include_base=True).cpu().numpy()
        one = be.circuit_probs(ansatz, n, L, tht[None]).cpu().numpy()
        be.set_option(dev, "alternate_walk", 1)
        alt = be.paramshift_probs(ansatz, n, L, tht, 0, P, include_base=True).cpu().numpy()
        one_alt = be.circuit_probs(ansatz, n, L, tht[None]).cpu().numpy()
    finally:
        be.set_option(dev, "alternate_walk", 1)
        be.set_option(dev, "prefix_share", 0)
    np.testing.assert_array_equal(alt, fwd)
    np.testing.assert_array_equal(one_alt, one)
    np.testing.assert_array_equal(one[0], fwd[0])
    np.testing.assert_allclose(fwd[0], oc.probs(ansatz, n, L, th), rtol=RTOL, atol=ATOL)
    t2 = th.copy(); t2[P - 1] -= np.pi / 2
    np.testing.assert_allclose(alt[2 * P], oc.probs(ansatz, n, L, t2), rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("ansatz", oc.ANSATZ_TYPES)
@pytest.mark.parametrize("n,L,kb", [(14, 3, 11), (15, 3, 13), (16, 2, 12)])
def test_read_map_planner_option(be, dev, ansatz, n, L, kb):
    """Planner option read_map (default: on with 8 amplitudes per thread, off with 16): phase-0 CNOTs may target thread-held wires; such a stage reads across
    thread groups and carries a barrier between its reads and its write-back (STAGE_CROSS_READ -- without it the rows
    were wrong on a cold GPU, where the waves of a workgroup drift apart).  Fewer stages, same circuit: a batch large
    enough for several tiles per workgroup against the oracle, and against the default plan to rounding."""
    be.set_option(dev, "tile_bits", kb)
    P = oc.num_params(ansatz, n, L)
    th = np.random.default_rng(3 * n + L).uniform(-np.pi, np.pi, P)
    tht = torch.as_tensor(th, device=dev)
    try:
        be.set_option(dev, "read_map", 0)
        ref = be.paramshift_probs(ansatz, n, L, tht, 0, P, include_base=True).clone()
        be.set_option(dev, "read_map", 1)
        got = be.paramshift_probs(ansatz, n, L, tht, 0, P, include_base=True)
    finally:
        be.set_option(dev, "read_map", -1)
    assert float((got.sum(dim=1) - 1).abs().max()) < 1e-12
    np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(got[0].cpu().numpy(), oc.probs(ansatz, n, L, th), rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("n,L,kb", [(8, 2, 13), (14, 2, 11)])
def test_strided_parameter_set(be, dev, n, L, kb):
    """bornvi_paramshift_probs_strided (the interleaved deal of a multi-GPU step): rows of rank r of W are the rows
    of parameters r, r + W, ... of the full batch, bit for bit, with and without the base row."""
    ansatz = "hardware_efficient"
    be.set_option(dev, "tile_bits", kb)
    P = oc.num_params(ansatz, n, L)
    tht = torch.as_tensor(np.random.default_rng(n).uniform(-1, 1, P), device=dev)
    full = be.paramshift_probs(ansatz, n, L, tht, 0, P, include_base=True).cpu().numpy()
    for W in (2, 3, 8):
        for r in (0, W - 1):
            mine = list(range(r, P, W))
            rows = [0] + [x for p_ in mine for x in (1 + 2 * p_, 2 + 2 * p_)]
            got = be.paramshift_probs(ansatz, n, L, tht, r, P, include_base=True, p_stride=W).cpu().numpy()
            np.testing.assert_array_equal(got, full[rows])
            got = be.paramshift_probs(ansatz, n, L, tht, r, P, include_base=False, p_stride=W).cpu().numpy()
            np.testing.assert_array_equal(got, full[rows[1:]])
    w = torch.as_tensor(np.random.default_rng(1).normal(size=2 ** n), device=dev)
    g_full = be.paramshift_grad(ansatz, n, L, tht, w, 0, P).cpu().numpy()
    g_str = be.paramshift_grad(ansatz, n, L, tht, w, 1, P, 4).cpu().numpy()
    np.testing.assert_array_equal(g_str, g_full[1::4])


def test_argument_errors_of_the_shift_entry_points(be, dev):
    """Out-of-range parameter sets, CU ranges and workspaces are refused with BORNVI_ERR_* (raised as BornviError), not
    executed: the kernels index with these numbers."""
    import ctypes as C
    from tensornetworks_amd import _ext
    ansatz, n, L = "hardware_efficient", 6, 2
    P = oc.num_params(ansatz, n, L)
    tht = torch.zeros(P, dtype=torch.float64, device=dev)
    h = _ext.handle_for(dev)
    aid = be.ansatz_id(ansatz)
    out = torch.empty((2 * P + 1, 1 << n), dtype=torch.float64, device=dev)
    ws = torch.empty(1 << 24, dtype=torch.uint8, device=dev)
    args = lambda p0, cnt, stride: ("bornvi_paramshift_probs_strided", aid, n, L, C.c_void_p(tht.data_ptr()), p0, cnt, stride, 1,
                                    C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(), None)
    h.call(*args(0, P, 1))                                    # the full set is fine
    for p0, cnt, stride in ((0, P + 1, 1), (-1, 1, 1), (0, 1, 0), (1, P, 1), (0, P // 2 + 1, 2), (P, 1, 1)):
        with pytest.raises(_ext.BornviError):
            h.call(*args(p0, cnt, stride))
    h.call(*args(P - 1, 1, 7))                                # last parameter alone, any stride
    with pytest.raises(_ext.BornviError):                     # workspace too small for one circuit is refused at n = 14
        be.set_option(dev, "tile_bits", 11)
        t14 = torch.zeros(oc.num_params(ansatz, 14, 1), dtype=torch.float64, device=dev)
        o14 = torch.empty((3, 1 << 14), dtype=torch.float64, device=dev)
        h.call("bornvi_paramshift_probs", aid, 14, 1, C.c_void_p(t14.data_ptr()), 0, 1, 1, C.c_void_p(o14.data_ptr()),
               C.c_void_p(ws.data_ptr()), 4096, None)
    st = C.c_void_p()
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    for first, cnt in ((-1, 4), (0, 0), (ncu - 1, 2), (0, ncu + 1)):
        with pytest.raises(_ext.BornviError):
            h.call("bornvi_stream_create_cu_range", first, cnt, C.byref(st))
    try:
        h.call("bornvi_stream_create_cu_range", 0, ncu // 2, C.byref(st))
    except _ext.BornviError as e:                             # a valid range: only the driver can refuse it
        pytest.skip(f"hipExtStreamCreateWithCUMask unavailable: {e}")
    assert st.value
    h.call("bornvi_stream_destroy", st)


def test_prefix_sharing_with_the_large_tile(be, dev):
    """n = 17 with the planner's own tile choice (2^13 tiles: one 512-thread workgroup per CU) -- shifted rows with
    and without prefix sharing are the same bits, and a few of them match the oracle."""
    ansatz, n, L = "hardware_efficient", 17, 2
    from tensornetworks_amd import _ext
    assert int(_ext.plan_words(_ext.ANSATZ_IDS[ansatz], n, L)[2]) == 13
    P = oc.num_params(ansatz, n, L)
    th = np.random.default_rng(17).uniform(-np.pi, np.pi, P)
    tht = torch.as_tensor(th, device=dev)
    try:
        be.set_option(dev, "prefix_share", 0)
        ref = be.paramshift_probs(ansatz, n, L, tht, 0, P, include_base=True)
        be.set_option(dev, "prefix_share", 1)
        got = be.paramshift_probs(ansatz, n, L, tht, 0, P, include_base=True)
        assert torch.equal(ref, got)
    finally:
        be.set_option(dev, "prefix_share", 0)
    got = got.cpu().numpy()
    np.testing.assert_allclose(got[0], oc.probs(ansatz, n, L, th), rtol=RTOL, atol=ATOL)
    t2 = th.copy(); t2[P - 1] -= np.pi / 2
    np.testing.assert_allclose(got[2 * P], oc.probs(ansatz, n, L, t2), rtol=RTOL, atol=ATOL)


def test_config4_full_depth_n20_L8_against_c_port(be, dev):
    """BASELINE config 4 at its REAL depth (n = 20, L = 8, hardware_efficient: P = 480, the planner's own 2^13-tile
    multi-pass plan with its fast tables; reference circuit quantum_born_machine.py:58-87).  Base row and the shifted
    rows of the first / last parameter and one in the middle, both signs, against the oracle's C port at 1e-9; every
    row sums to 1; the full 961-row batch with prefix sharing on is bitwise the batch with it off."""
    from oracle import cpu_port as cp
    if not cp.available():
        pytest.skip("oracle/_build/libcpu_port.so not built")
    ansatz, n, L = "hardware_efficient", 20, 8
    P = oc.num_params(ansatz, n, L)
    assert P == 480
    g = torch.Generator().manual_seed(0)
    th = (0.1 * torch.randn(P, generator=g, dtype=torch.float32)).double().numpy()      # bench.py's theta0
    tht = torch.as_tensor(th, device=dev)
    picks = [0, P // 2 + 7, P - 1]
    want_t = [th.copy()]
    for p_ in picks:
        for sgn in (+1, -1):
            t2 = th.copy(); t2[p_] += sgn * np.pi / 2
            want_t.append(t2)
    want = cp.circuit_probs(ansatz, n, L, np.stack(want_t))
    try:
        be.set_option(dev, "prefix_share", 0)
        full = be.paramshift_probs(ansatz, n, L, tht, 0, P, include_base=True)
        assert full.shape == (2 * P + 1, 1 << n)
        sums = full.sum(dim=1)
        assert float((sums - 1.0).abs().max()) < 1e-12
        rows = [0] + [r for p_ in picks for r in (1 + 2 * p_, 2 + 2 * p_)]
        got = full[rows].cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-17)
        # the single-parameter entry point gives the same rows (what one rank of W evaluates)
        one = be.paramshift_probs(ansatz, n, L, tht, P - 1, P, include_base=False)
        assert torch.equal(one, full[2 * P - 1:])
        be.set_option(dev, "prefix_share", 1)
        shared = be.paramshift_probs(ansatz, n, L, tht, 0, P, include_base=True)
        assert torch.equal(shared, full)
    finally:
        be.set_option(dev, "prefix_share", 0)
        be.release_workspaces()
