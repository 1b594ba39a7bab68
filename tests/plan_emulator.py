"""NumPy emulator of the circuit execution plan (test infrastructure only).

Interprets the uint32 plan produced by the C++ planner (bornvi_plan_describe) exactly as
circuit_pass_kernel does -- same tile/bit-permutation arithmetic, same stage/thread/register
decomposition, same micro-ops -- but on NumPy arrays, so that the planner can be validated
against the oracle without a GPU.  The LDS swizzle is emulated too: the tile array is indexed by LOGICAL index, the swizzled
addresses the kernel forms are mapped back through the (involutive) swizzle.
"""
import numpy as np

from oracle import circuit as oc

PH_N, PH_K, PH_NPASSES, PH_NFUSED, PH_NPARAMS, PH_OFF_FUSED, PH_OFF_PASSTAB, PH_TOTAL, PH_THREADS, PH_R, PH_NGATES = range(1, 12)
FUSED_WORDS = 10
PW_FLAGS, PW_K, PW_N, PW_NSTAGES, PW_LO_IN, PW_LO_OUT, PW_THREADS = range(7)
PW_IN_PHYS, PW_IN_GPHYS, PW_OUT_PHYS, PW_OUT_GPHYS, PW_IN_MASK, PW_IN_GMASK, PW_OUT_MASK, PW_OUT_GMASK = 8, 12, 16, 20, 24, 32, 40, 48
PW_WIRE_OF_LDS, PW_WIRE_OF_G, PW_MATS, PW_OUT_COL, PW_OUT_GCOL, PW_STAGES = 64, 96, 128, 192, 208, 224
PASS_INIT, PASS_FINAL, PASS_FINAL_STATE = 1, 2, 4
STAGE_HDR_WORDS = 64
STAGE_CROSS_READ = 16
STAGE_SIGN_PRE, STAGE_SIGN_POST = 1, 2
SIGNQ_WORDS = 49
KIND_NAMES = {0: "H", 1: "RX", 2: "RY", 3: "RZ"}
# fast-path tables (plan.hpp: FastHeader / FastStage)
FH_NSTAGES, FH_RW_BASE, FH_SG_BASE, FH_SIGN_PRE, FH_SIGN_POST, FH_IN_TAB, FH_OUT_TAB = 0, 1, 2, 3, 4, 5, 6
FH_IN_BASIS, FH_OUT_BASIS, FH_WORDS = 8, 12, 16
FH_ZINFO = 7      # support of |0..0>: INIT pass = mask over tile-index bits of the tiles nobody reads; next pass = zero slots
STAGE_FROM_HBM, STAGE_TO_HBM = 4, 8
FS_FI01, FS_FI23, FS_RB, FS_WB, FS_KIND, FS_WORDS = 0, 1, 2, 6, 10, 16


def fused_matrices(W, theta):
    nf = int(W[PH_NFUSED]); off = int(W[PH_OFF_FUSED])
    mats = []
    for f in range(nf):
        fw = W[off + f * FUSED_WORDS: off + (f + 1) * FUSED_WORDS]
        U = np.eye(2, dtype=np.complex128)
        for e in range(int(fw[1])):
            kind, par = int(fw[2 + 2 * e]), int(fw[3 + 2 * e])
            t = None if par == 0xFFFFFFFF else theta[par]
            U = oc.matrix_1q(KIND_NAMES[kind], t) @ U
        mats.append(U)
    return mats


def tbyte(P, table, j):
    """Entry j of a byte-packed 16-entry table of the pass header."""
    return (int(P[table + (j >> 2)]) >> (8 * (j & 3))) & 0xFF


def thalf(P, table, j):
    """Entry j of a 16-bit-packed 16-entry table of the pass header."""
    return (int(P[table + (j >> 1)]) >> (16 * (j & 1))) & 0xFFFF


def xor_map(v, nbits, P, table):
    o = np.zeros_like(v)
    for j in range(nbits):
        o ^= ((v >> j) & 1) * thalf(P, table, j)
    return o


def swz(l):
    """lds_swizzle of plan.hpp (linear over GF(2), an involution)."""
    return l ^ (((l >> 4) ^ (l >> 8) ^ (l >> 12)) & 15)


swz_inv = swz


def apply_sign(Q, e, n, amp, nreg):
    acc = np.zeros(e.shape, dtype=np.uint64)
    for q in range(n):
        acc ^= ((e >> q) & 1).astype(np.uint64) & popc(e & int(Q[q]))
    qbits = int(Q[48])
    for j in range(nreg):
        sgn = (((qbits >> j) & 1) ^ (acc & np.uint64(1)) ^ (popc(e & int(Q[32 + j])) & np.uint64(1))).astype(bool)
        amp[j] = np.where(sgn, -amp[j], amp[j])


def popc(x):
    x = np.asarray(x, dtype=np.uint64)
    c = np.zeros(x.shape, dtype=np.uint64)
    for s in range(32):
        c += (x >> np.uint64(s)) & np.uint64(1)
    return c


def fast_stage(F, FH, s, g, k, n, tile, mats, direct_in=None, direct_out=None, zslots=0, r=4):
    """One stage exactly as circuit_pass_fast_kernel runs it: slots and signs from the planner tables.
    direct_in = (buf, lo_in): the first stage takes its amplitudes straight from the pass's input buffer at the byte
    offsets of the FH_IN_TAB table (the tile, filled the ordinary way, must hold the same values);
    direct_out = (phys_of_slot, shift): the last stage's HBM offsets (FH_OUT_TAB) must be where the ordinary tile
    drain would put each slot."""
    kt = k - r
    nthr = 1 << kt
    nslots = 1 << r
    FS = FH + FH_WORDS + s * FS_WORDS
    per_stage = 1 << (n - r)
    rw = F[int(F[FH + FH_RW_BASE]) + s * per_stage + (g << kt): int(F[FH + FH_RW_BASE]) + s * per_stage + (g << kt) + nthr].astype(np.int64)
    pre = (int(F[FH + FH_SIGN_PRE]) >> s) & 1
    post = (int(F[FH + FH_SIGN_POST]) >> s) & 1
    kind = int(F[FS + FS_KIND])
    assert ((kind >> 3) & 1) == pre and ((kind >> 4) & 1) == post
    ng = kind & 7
    fi = [int(F[FS + FS_FI01]) & 0xFFFF, int(F[FS + FS_FI01]) >> 16, int(F[FS + FS_FI23]) & 0xFFFF, int(F[FS + FS_FI23]) >> 16]
    assert all(f != 0xFFFF for f in fi[:ng]) and all(f == 0xFFFF for f in fi[ng:])
    rb = [int(F[FS + FS_RB + b]) for b in range(r)]
    wb = [int(F[FS + FS_WB + b]) for b in range(r)]

    def addr(base_slot, basis, j):
        a = base_slot << 4
        for b in range(r):
            if (j >> b) & 1:
                a = a ^ basis[b]
        assert np.all(a % 16 == 0)
        return swz_inv(a >> 4)
    rd = [addr(rw & 0xFFFF, rb, j) for j in range(nslots)]
    wr = [addr(rw >> 16, wb, j) for j in range(nslots)]
    assert np.array_equal(np.sort(np.concatenate(rd)), np.arange(1 << k))
    assert np.array_equal(np.sort(np.concatenate(wr)), np.arange(1 << k))
    # (the write-back stays in the thread's own group of 16 slots -- the slots differing in the register positions only;
    # the reads may come from anywhere in the tile: nothing has been written in this stage yet)
    own0 = np.stack(wr)
    regmask = 0
    for b in range(r):
        regmask |= int(swz_inv(np.array([wb[b] >> 4]))[0])
    assert np.all((own0 & ~regmask) == (own0[0] & ~regmask)[None, :])
    amp = [tile[rd[j]].copy() for j in range(nslots)]
    if direct_in is not None:
        buf, lo_in = direct_in
        tab = int(F[FH + FH_IN_TAB])
        off0 = F[tab + (g << kt): tab + (g << kt) + nthr].astype(np.int64)
        basis = [int(F[FH + FH_IN_BASIS + b]) for b in range(r)]
        for j in range(nslots):
            off = off0.copy()
            for b in range(r):
                if (j >> b) & 1:
                    off ^= basis[b]
            assert np.all(off % 16 == 0)
            if (zslots >> j) & 1:
                # known zero (support of |0..0>): the kernel does not load this slot -- what lies there is either the
                # zeros pass 0 wrote or the poison of a tile it left out; the stage computes with 0
                assert np.all((buf[off >> 4] == 0) | np.isnan(buf[off >> 4]))
                amp[j] = np.zeros_like(amp[j])
                continue
            np.testing.assert_array_equal(buf[off >> 4], amp[j])      # same amplitudes as through the tile
            lanes = 1 << min(lo_in, 6)
            for w0 in range(0, nthr, lanes):                          # each group of 2^lo_in lanes loads one aligned run
                run = np.sort(off[w0:w0 + lanes])
                assert np.array_equal(run, (run[0] & ~(16 * lanes - 1)) + 16 * np.arange(lanes))
    if pre or post:
        sgw = F[int(F[FH + FH_SG_BASE]) + s * per_stage + (g << kt): int(F[FH + FH_SG_BASE]) + s * per_stage + (g << kt) + nthr].astype(np.int64)
    if pre:
        for j in range(nslots):
            amp[j] = np.where(((sgw >> j) & 1).astype(bool), -amp[j], amp[j])
    for a in range(ng):
        U = mats[fi[a]]
        for j in range(nslots):
            if j & (1 << a):
                continue
            j1 = j | (1 << a)
            x0, x1 = amp[j], amp[j1]
            amp[j], amp[j1] = U[0, 0] * x0 + U[0, 1] * x1, U[1, 0] * x0 + U[1, 1] * x1
    if post:
        for j in range(nslots):
            amp[j] = np.where(((sgw >> (16 + j)) & 1).astype(bool), -amp[j], amp[j])
    for j in range(nslots):
        tile[wr[j]] = amp[j]
    if direct_out is not None:
        phys_of_slot, sh = direct_out
        tab = int(F[FH + FH_OUT_TAB])
        off0 = F[tab + (g << kt): tab + (g << kt) + nthr].astype(np.int64)
        basis = [int(F[FH + FH_OUT_BASIS + b]) for b in range(r)]
        for j in range(nslots):
            off = off0.copy()
            for b in range(r):
                if (j >> b) & 1:
                    off ^= basis[b]
            assert np.all(off % (1 << sh) == 0)
            np.testing.assert_array_equal(off >> sh, phys_of_slot[wr[j]])    # where the tile drain would write the slot


def run_plan(W, mats, state_in=None, fast=None):
    """Returns probabilities (PASS_FINAL) or the final state (PASS_FINAL_STATE), canonical order.
    fast = (words, pass offsets) of the fast-path tables: the stages are then taken from those tables (as
    circuit_pass_fast_kernel does) instead of the stage headers (as circuit_pass_kernel does)."""
    W = np.asarray(W, dtype=np.uint32)
    n, k, np_ = int(W[PH_N]), int(W[PH_K]), int(W[PH_NPASSES])
    N = 1 << n
    buf = None if state_in is None else np.asarray(state_in, dtype=np.complex128).copy()
    result = None
    for pi in range(np_):
        P = W[int(W[int(W[PH_OFF_PASSTAB]) + pi]):]
        flags = int(P[PW_FLAGS]); T = int(P[PW_THREADS]); nst = int(P[PW_NSTAGES])
        lo_in, lo_out = int(P[PW_LO_IN]), int(P[PW_LO_OUT])
        assert int(P[PW_K]) == k and int(P[PW_N]) == n
        ksize = 1 << k
        out = np.zeros(N, dtype=np.complex128)
        probs = np.zeros(N)
        written = np.zeros(N, dtype=np.int64)
        # structural checks the kernel relies on
        for j in range(lo_in):
            assert (flags & PASS_INIT) or tbyte(P, PW_IN_PHYS, j) == j
        for j in range(lo_out):
            assert (flags & PASS_FINAL) or tbyte(P, PW_OUT_PHYS, j) == j
        for g in range(1 << (n - k)):
            u = np.arange(ksize, dtype=np.int64)
            if flags & PASS_INIT:
                tile = np.zeros(ksize, dtype=np.complex128)
                if g == 0:
                    tile[0] = 1.0
            else:
                phys = np.zeros(ksize, dtype=np.int64)
                for j in range(k):
                    phys |= ((u >> j) & 1) << tbyte(P, PW_IN_PHYS, j)
                for m in range(n - k):
                    phys |= ((g >> m) & 1) << tbyte(P, PW_IN_GPHYS, m)
                # LDS slot of each loaded element: head CNOTs of the pass are folded into the slot masks
                gvec = np.full(ksize, g, dtype=np.int64)
                slot = swz_inv(xor_map(u, k, P, PW_IN_MASK) ^ xor_map(gvec, n - k, P, PW_IN_GMASK))
                assert np.array_equal(np.sort(slot), np.arange(ksize))
                tile = np.zeros(ksize, dtype=np.complex128)
                tile[slot] = buf[phys]
            S = P[PW_STAGES:]
            if fast is not None:
                Fw, Foffs = fast
                FH = int(Foffs[pi])
                assert int(Fw[FH + FH_NSTAGES]) == nst
                # where the ordinary tile drain writes each (logical) tile slot
                vv = np.arange(ksize, dtype=np.int64)
                gv = np.full(ksize, g, dtype=np.int64)
                lds_v = swz_inv(xor_map(vv, k, P, PW_OUT_MASK) ^ xor_map(gv, n - k, P, PW_OUT_GMASK))
                phys_v = np.zeros(ksize, dtype=np.int64)
                for j in range(k):
                    phys_v ^= ((vv >> j) & 1) * int(P[PW_OUT_COL + j])
                for m in range(n - k):
                    phys_v ^= ((g >> m) & 1) * int(P[PW_OUT_GCOL + m])
                phys_of_slot = np.zeros(ksize, dtype=np.int64)
                phys_of_slot[lds_v] = phys_v
                use_in = int(Fw[FH + FH_IN_TAB]) != 0 and not (flags & PASS_INIT) and nst > 0
                use_out = int(Fw[FH + FH_OUT_TAB]) != 0 and nst > 1
                zslots = int(Fw[FH + FH_ZINFO]) & 0xFFFF if (use_in and not (flags & PASS_INIT)) else 0
                for si in range(nst):
                    fast_stage(Fw, FH, si, g, k, n, tile, mats, zslots=zslots if si == 0 else 0, r=int(W[PH_R]),
                               direct_in=(buf, lo_in) if (si == 0 and use_in) else None,
                               direct_out=(phys_of_slot, 3 if (flags & PASS_FINAL) else 4) if (si == nst - 1 and use_out) else None)
            for si in range(nst if fast is None else 0):
                # the per-pass matrix table (what the kernel stages into LDS) must agree with the stage header
                assert int(P[PW_MATS + 2 * si]) == int(S[6]) and int(P[PW_MATS + 2 * si + 1]) == int(S[7])
                hdr = int(S[0]); r = hdr & 0xFF; sflags = (hdr >> 8) & 0xFF; nwords = hdr >> 16
                rho = int(S[1])
                nthr = 1 << (k - r)
                assert nthr <= T
                t = np.arange(nthr, dtype=np.int64)
                base = np.zeros(nthr, dtype=np.int64)
                for j in range(k - r):
                    pos = (int(S[2 + (j >> 2)]) >> (8 * (j & 3))) & 0xFF
                    base |= ((t >> j) & 1) << pos
                rpos = [(rho >> (8 * i)) & 0xFF for i in range(r)]
                nreg = 1 << r
                fi = [int(S[6]) & 0xFFFF, int(S[6]) >> 16, int(S[7]) & 0xFFFF, int(S[7]) >> 16]
                e = base | (g << k)
                pb = swz(base)
                lflip = np.zeros(nthr, dtype=np.int64); sflip = np.zeros(nthr, dtype=np.int64); e2 = e.copy()
                for i in range(r):
                    sri = swz(1 << rpos[i])
                    bpre, bpost = int(S[8 + i]), int(S[12 + i])
                    lflip ^= (popc(e & bpre) & np.uint64(1)).astype(np.int64) * sri
                    bit = (popc(e & bpost) & np.uint64(1)).astype(np.int64)
                    sflip ^= bit * sri
                    e2 |= bit << rpos[i]
                for i in range(r, 4):
                    assert int(S[8 + i]) == 0 and int(S[12 + i]) == 0
                for p2 in range(16):                       # BpreF: the read map moves thread-held positions too
                    m = int(S[48 + p2])
                    assert m == 0 or (p2 < k and p2 not in rpos)
                    if m:
                        lflip ^= (popc(e & m) & np.uint64(1)).astype(np.int64) * swz(1 << p2)
                loff = [int(S[16 + j]) for j in range(16)]
                soff = [int(S[32 + j]) for j in range(16)]
                rd = [swz_inv(pb ^ lflip ^ loff[j]) for j in range(nreg)]
                wr = [swz_inv(pb ^ sflip ^ soff[j]) for j in range(nreg)]
                # a thread reads and writes exactly its own 2^r-element group (no cross-thread hazard),
                # and the groups partition the tile
                own = np.sort(np.stack([base ^ sum(((j >> i) & 1) << rpos[i] for i in range(r)) for j in range(nreg)]), axis=0)
                # (reads may come from other threads' groups: the stage then carries STAGE_CROSS_READ, and the kernels put a
                # barrier between its reads and its write-back, which stays inside the thread's own group)
                assert np.array_equal(np.sort(np.stack(wr), axis=0), own)
                crosses = not np.array_equal(np.sort(np.stack(rd), axis=0), own)
                assert bool(sflags & STAGE_CROSS_READ) or not crosses
                assert np.array_equal(np.sort(np.concatenate(rd)), np.arange(ksize))
                amp = [tile[rd[j]].copy() for j in range(nreg)] + [np.zeros(nthr, np.complex128)] * (16 - nreg)
                q_off = STAGE_HDR_WORDS
                if sflags & STAGE_SIGN_PRE:
                    apply_sign(S[q_off: q_off + SIGNQ_WORDS], e, n, amp, nreg)
                    q_off += SIGNQ_WORDS
                for a in range(4):
                    if fi[a] == 0xFFFF:
                        continue
                    assert a < r
                    U = mats[fi[a]]
                    for j in range(nreg):
                        if j & (1 << a):
                            continue
                        j1 = j | (1 << a)
                        x0, x1 = amp[j], amp[j1]
                        amp[j], amp[j1] = U[0, 0] * x0 + U[0, 1] * x1, U[1, 0] * x0 + U[1, 1] * x1
                if sflags & STAGE_SIGN_POST:
                    apply_sign(S[q_off: q_off + SIGNQ_WORDS], e2, n, amp, nreg)
                    q_off += SIGNQ_WORDS
                # per-thread base table emitted by the planner == the deposit the header describes
                tab = S[q_off: q_off + nthr].astype(np.int64)
                assert np.array_equal(tab & 0xFFFF, pb) and np.array_equal(tab >> 16, base)
                q_off += nthr
                assert q_off == nwords, (q_off, nwords)
                for j in range(nreg):
                    tile[wr[j]] = amp[j]
                S = S[nwords:]
            # store
            v = np.arange(ksize, dtype=np.int64)
            gvec = np.full(ksize, g, dtype=np.int64)
            lds = swz_inv(xor_map(v, k, P, PW_OUT_MASK) ^ xor_map(gvec, n - k, P, PW_OUT_GMASK))   # tail CNOTs folded in
            # phys-out address: GF(2)-linear columns; one-hot (1 << OUT_PHYS) except where the circuit-ending
            # CNOTs are applied to the outcome index of the probabilities (last pass)
            phys = np.zeros(ksize, dtype=np.int64)
            for j in range(k):
                phys ^= ((v >> j) & 1) * int(P[PW_OUT_COL + j])
            for m in range(n - k):
                phys ^= ((g >> m) & 1) * int(P[PW_OUT_GCOL + m])
            if not (flags & PASS_FINAL):
                for j in range(k):
                    assert int(P[PW_OUT_COL + j]) == 1 << tbyte(P, PW_OUT_PHYS, j)
                for m in range(n - k):
                    assert int(P[PW_OUT_GCOL + m]) == 1 << tbyte(P, PW_OUT_GPHYS, m)
            # a 64-lane store covers one aligned run of 2^lo_out elements; with the circuit-ending CNOT ring in the
            # address (its wrap-around CNOT puts the lanes' parity into the top bit) it splits over two runs whose
            # other halves the neighbouring store instructions of the same wave fill
            lanes = phys[: 1 << lo_out]
            lowmask = (1 << lo_out) - 1
            assert np.array_equal(np.sort(lanes & lowmask), np.arange(1 << lo_out))
            assert len(np.unique(lanes & ~lowmask)) <= (2 if (flags & PASS_FINAL) else 1)
            written[phys] += 1
            assert np.array_equal(np.sort(lds), np.arange(ksize))
            if flags & PASS_FINAL:
                probs[phys] = np.abs(tile[lds]) ** 2
            elif fast is not None and (flags & PASS_INIT) and (g & int(fast[0][int(fast[1][pi]) + FH_ZINFO])):
                assert not np.any(tile)                    # a tile the INIT pass leaves out is all zero ...
                out[phys] = np.nan                         # ... and nobody may read what lies there instead
            else:
                out[phys] = tile[lds]
        assert np.all(written == 1)      # the phys-out map is a bijection
        buf = out
        if flags & PASS_FINAL:
            result = probs
        elif flags & PASS_FINAL_STATE:
            result = out
    return result


def plan_stats(W):
    n, k, np_ = int(W[PH_N]), int(W[PH_K]), int(W[PH_NPASSES])
    stages = []
    for pi in range(np_):
        P = W[int(W[int(W[PH_OFF_PASSTAB]) + pi]):]
        stages.append(int(P[PW_NSTAGES]))
    return {"n": n, "k": k, "passes": np_, "stages": stages, "fused": int(W[PH_NFUSED]), "gates": int(W[PH_NGATES])}


# ---- compact tables (plan.hpp: CompactTables; kernels_circuit8.hip: circuit_pass_r3_kernel) -----------------------
CH_NSTAGES, CH_NROWS, CH_NSIGN, CH_SIGN_PRE, CH_SIGN_POST, CH_DIRECT, CH_ZINFO, CH_NWAVES, CH_MAT_OFF, CH_LANE_OFF, CH_UNI_OFF, CH_MASK_OFF = range(12)
CH_IN_STEP_D, CH_IN_STEP_N, CH_FILL_STEP, CH_DRAIN_STEP, CH_OUT_STEP_D, CH_OUT_STEP_N, CH_WORDS = 12, 15, 18, 21, 24, 27, 32
CS_KIND, CS_CROSS, CS_RB, CS_WB, CS_MAT, CS_WORDS = 0, 1, 2, 5, 8, 12
CH_NMAT = 30
CR_IN_D, CR_IN_N, CR_OUT_D, CR_OUT_N, CR_SLOT = range(5)


def _comb3(j, basis):
    return (basis[0] if j & 1 else 0) ^ (basis[1] if j & 2 else 0) ^ (basis[2] if j & 4 else 0)


def run_plan_compact(W, compact, mats, state_in=None, direct=3, zero_support=True):
    """The plan with 3 register wires run exactly as circuit_pass_r3_kernel runs it: every per-thread word is
    LANE[row][lane] ^ UNI[tile row][row][wave] (sign rows: the bilinear lane x (wave, tile row) term through MASK), the tile is
    filled / drained through rows CR_IN_N / CR_SLOT / CR_OUT_N or, where the tables allow, the first / last stage moves its
    amplitudes straight between the buffers and the registers (rows CR_IN_D / CR_OUT_D)."""
    W = np.asarray(W, dtype=np.uint32)
    Cw, Coffs = compact
    Cw = np.asarray(Cw, dtype=np.int64)
    n, k, np_ = int(W[PH_N]), int(W[PH_K]), int(W[PH_NPASSES])
    assert int(W[PH_R]) == 3
    N, ksize, kt, gbits = 1 << n, 1 << k, k - 3, n - k
    T = 1 << kt
    t = np.arange(T, dtype=np.int64)
    lane, wv = t & 63, t >> 6
    buf = None if state_in is None else np.asarray(state_in, dtype=np.complex128).copy()
    result = None
    skipped = set()
    for pi in range(np_):
        P = W[int(W[int(W[PH_OFF_PASSTAB]) + pi]):]
        flags = int(P[PW_FLAGS])
        C = int(Coffs[pi])
        H = [int(x) for x in Cw[C: C + CH_WORDS]]
        nst, nrows, nsign, NW = H[CH_NSTAGES], H[CH_NROWS], H[CH_NSIGN], H[CH_NWAVES]
        assert nst == int(P[PW_NSTAGES]) and NW == max(T // 64, 1) and int(P[PW_THREADS]) == max(64, T)
        sign_any = H[CH_SIGN_PRE] | H[CH_SIGN_POST]
        init, fin = bool(flags & PASS_INIT), bool(flags & PASS_FINAL)
        out_shift = 3 if fin else 4
        direct_in = bool(H[CH_DIRECT] & 1) and not init and nst > 0 and bool(direct & 1)
        direct_out = bool(H[CH_DIRECT] & 2) and nst > 1 and bool(direct & 2)
        zs_ok = zero_support and direct == 3 and state_in is None
        zinfo = H[CH_ZINFO] if zs_ok else 0
        zgmask = zinfo if (init and gbits > 0) else 0
        zslots = (zinfo & 0xFF) if (not init and direct_in) else 0
        row0 = nst + nsign

        def word(row, g):
            lw = Cw[C + H[CH_LANE_OFF] + row * 64 + lane]
            uw = Cw[C + H[CH_UNI_OFF] + (g * nrows + row) * NW + wv]
            return lw ^ uw

        def sign_word(sr, g):
            w_ = word(nst + sr, g)
            mk = Cw[C + H[CH_MASK_OFF] + (g * nsign + sr) * NW + wv]
            ppre = (popc(lane & (mk & 0xFF)) & np.uint64(1)).astype(np.int64)
            ppost = (popc(lane & (mk >> 8)) & np.uint64(1)).astype(np.int64)
            return w_ ^ (ppre * 0xFFFF) ^ (ppost * 0xFFFF0000)

        steps = {name: [H[base + m] for m in range(3)] for name, base in
                 (("in_d", CH_IN_STEP_D), ("in_n", CH_IN_STEP_N), ("fill", CH_FILL_STEP), ("drain", CH_DRAIN_STEP),
                  ("out_d", CH_OUT_STEP_D), ("out_n", CH_OUT_STEP_N))}
        # the fused matrix of (stage, register bit): byte offset in one circuit's gate array (64 bytes per matrix); the
        # pass's list of matrices (touched a trip ahead) holds exactly those
        mat_of, listed = {}, []
        for s_ in range(nst):
            ng_ = int(Cw[C + CH_WORDS + s_ * CS_WORDS + CS_KIND]) & 7
            for gi in range(ng_):
                off = int(Cw[C + CH_WORDS + s_ * CS_WORDS + CS_MAT + gi])
                assert off % 64 == 0
                mat_of[(s_, gi)] = off >> 6
                listed.append(off)
        assert listed == [int(x) for x in Cw[C + H[CH_MAT_OFF]: C + H[CH_MAT_OFF] + int(Cw[C + CH_NMAT])]]
        out = np.zeros(N, dtype=np.complex128)
        probs = np.zeros(N)
        written = np.zeros(N, dtype=np.int64)
        for g in range(1 << gbits):
            zero_tile = init and gbits > 0 and g != 0
            if zero_tile and (g & zgmask):
                skipped.add((pi, g))
                continue                          # nobody reads this tile: not written at all (stays poison below)
            tile = np.zeros(ksize, dtype=np.complex128)          # indexed by SWIZZLED slot, as the kernel's LDS is
            amp = None
            if init:
                if g == 0:
                    tile[0] = 1.0
            elif direct_in:
                base = word(row0 + CR_IN_D, g)
                amp = []
                for j in range(8):
                    off = base ^ _comb3(j, steps["in_d"])
                    assert np.all(off % 16 == 0)
                    if (zslots >> j) & 1:
                        assert np.all((buf[off >> 4] == 0) | np.isnan(buf[off >> 4]))
                        amp.append(np.zeros(T, dtype=np.complex128))
                    else:
                        assert not np.any(np.isnan(buf[off >> 4]))
                        amp.append(buf[off >> 4].copy())
            else:
                base = word(row0 + CR_IN_N, g)
                slot = word(row0 + CR_SLOT, g) & 0xFFFF
                seen = np.zeros(ksize, dtype=np.int64)
                for i in range(8):
                    off = base ^ _comb3(i, steps["in_n"])
                    sl = slot ^ _comb3(i, steps["fill"])
                    tile[sl] = buf[off >> 4]
                    seen[sl] += 1
                assert np.all(seen == 1)
            for s in range(0 if not zero_tile else nst, nst):
                CS = C + CH_WORDS + s * CS_WORDS
                kind = int(Cw[CS + CS_KIND])
                ng, pre, post = kind & 7, (kind >> 3) & 1, (kind >> 4) & 1
                assert ng <= 3 and int(Cw[CS + CS_CROSS]) == 0, "the sequential emulator cannot order a cross-group read"
                rb = [int(Cw[CS + CS_RB + b]) for b in range(3)]
                wb = [int(Cw[CS + CS_WB + b]) for b in range(3)]
                rw = word(s, g)
                if s == 0 and direct_in:
                    a = amp
                else:
                    ra0 = (rw & 0xFFFF) << 4
                    rd = [(ra0 ^ _comb3(j, rb)) >> 4 for j in range(8)]
                    assert np.array_equal(np.sort(np.concatenate(rd)), np.arange(ksize))
                    a = [tile[rd[j]].copy() for j in range(8)]
                if pre or post:
                    sg = sign_word(bin(sign_any & ((1 << s) - 1)).count("1"), g)
                if pre:
                    a = [np.where(((sg >> j) & 1).astype(bool), -a[j], a[j]) for j in range(8)]
                for gi in range(ng):
                    U = mats[mat_of[(s, gi)]]
                    for j in range(8):
                        if j & (1 << gi):
                            continue
                        j1 = j | (1 << gi)
                        x0, x1 = a[j], a[j1]
                        a[j], a[j1] = U[0, 0] * x0 + U[0, 1] * x1, U[1, 0] * x0 + U[1, 1] * x1
                if post:
                    a = [np.where(((sg >> (16 + j)) & 1).astype(bool), -a[j], a[j]) for j in range(8)]
                if s == nst - 1 and direct_out:
                    base = word(row0 + CR_OUT_D, g)
                    for j in range(8):
                        off = base ^ _comb3(j, steps["out_d"])
                        assert np.all(off % (1 << out_shift) == 0)
                        written[off >> out_shift] += 1
                        if fin:
                            probs[off >> out_shift] = np.abs(a[j]) ** 2
                        else:
                            out[off >> out_shift] = a[j]
                else:
                    wa0 = (rw >> 16) << 4
                    wr = [(wa0 ^ _comb3(j, wb)) >> 4 for j in range(8)]
                    assert np.array_equal(np.sort(np.concatenate(wr)), np.arange(ksize))
                    for j in range(8):
                        tile[wr[j]] = a[j]
            if zero_tile or not direct_out:
                if zero_tile:      # the kernel forms these addresses from the pass header (no table rows for g != 0)
                    off0 = np.zeros(T, dtype=np.int64)
                    for j in range(kt):
                        off0 ^= ((t >> j) & 1) * int(P[PW_OUT_COL + j])
                    for m in range(gbits):
                        off0 ^= ((g >> m) & 1) * int(P[PW_OUT_GCOL + m])
                    off0 = off0 << out_shift
                else:
                    off0 = word(row0 + CR_OUT_N, g)
                slot = word(row0 + CR_SLOT, g) >> 16 if not zero_tile else np.zeros(T, dtype=np.int64)
                for i in range(8):
                    off = off0 ^ _comb3(i, steps["out_n"])
                    x = tile[slot ^ _comb3(i, steps["drain"])] if not zero_tile else np.zeros(T, dtype=np.complex128)
                    written[off >> out_shift] += 1
                    if fin:
                        probs[off >> out_shift] = np.abs(x) ** 2
                    else:
                        out[off >> out_shift] = x
        if skipped and init:
            assert np.all(written <= 1)
            out[written == 0] = np.nan            # tiles the INIT pass leaves out: nobody may read what lies there
        else:
            assert np.all(written == 1)
        buf = out
        if fin:
            result = probs
        elif flags & PASS_FINAL_STATE:
            result = out
    return result
