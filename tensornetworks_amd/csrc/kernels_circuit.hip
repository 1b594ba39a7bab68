// Statevector kernels for gfx950 (MI355X): fused-gate matrices, the LDS-tiled circuit pass,
// the un-fused one-gate kernels (gate-apply micro-benchmark) and Born probabilities.
//
// Data layout: a state is 2^n complex128 (re, im interleaved = double2), outcome index i with
// wire 0 the MOST significant bit (utils.py:77-91 / qml.probs order).  Between passes the
// amplitudes live in HBM in a pass-specific bit permutation chosen by the planner (plan.hpp);
// only the last pass writes the canonical order (as probabilities).
#include <hip/hip_runtime.h>

#include "circuit_dev.hpp"
#include "kernels.hpp"
#include "plan.hpp"

namespace bornvi {

// LDS index swizzle `lds_swizzle` (plan.hpp): with it a 16-lane ds_read_b128 group is conflict-free
// whenever its lane bits land on LDS bit positions with distinct residues mod 4 (the planner's
// order_for_banks), including register wires on bits 0-3.

// ------------------------------------------------------------------------------------------------
// fused one-qubit matrices: U = G_{ne-1} ... G_1 G_0 (G_0 applied first), PennyLane conventions
// RX = exp(-i t X/2), RY = exp(-i t Y/2), RZ = diag(e^{-it/2}, e^{it/2}); H.
// Row-major complex: m[0..1] = u00, m[2..3] = u01, m[4..5] = u10, m[6..7] = u11.
// ------------------------------------------------------------------------------------------------
__device__ inline void elem_matrix(uint32_t kind, double t, double (&m)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) m[i] = 0.0;
  if (kind == G_H) {
    const double h = 0.70710678118654752440;
    m[0] = h; m[2] = h; m[4] = h; m[6] = -h;
    return;
  }
  double s, c;
  sincos(t / 2.0, &s, &c);
  if (kind == G_RX) { m[0] = c; m[3] = -s; m[5] = -s; m[6] = c; }
  else if (kind == G_RY) { m[0] = c; m[2] = -s; m[4] = s; m[6] = c; }
  else { m[0] = c; m[1] = -s; m[6] = c; m[7] = s; }  // RZ
}

__global__ void build_gates_kernel(const uint32_t* __restrict__ plan, const double* __restrict__ thetas,
                                   long long theta_stride, int shift_mode, int p_begin, int p_stride, int include_base,
                                   long long b_offset, int batch, double* __restrict__ gates,
                                   const int* __restrict__ shift_tab, int slots, int normalise) {
  const uint32_t nf = plan[PH_NFUSED];
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)batch * nf) return;
  const long long b = idx / nf;
  const uint32_t f = (uint32_t)(idx % nf);
  const uint32_t* fw = plan + plan[PH_OFF_FUSED] + f * FUSED_WORDS;
  const long long bg = b + b_offset;
  uint32_t pshift = 0xfffffffeu;
  double shift = 0.0;
  if (shift_tab) {                       // circuit b of this launch: parameter * 2 + (1 for the minus shift), < 0: base
    const int code = shift_tab[b];
    if (code >= 0) { pshift = (uint32_t)(code >> 1); shift = (code & 1) ? -M_PI_2 : M_PI_2; }
  } else if (shift_mode) {
    const long long bb = bg - include_base;
    if (bb >= 0) { pshift = (uint32_t)(p_begin + (bb >> 1) * p_stride); shift = (bb & 1) ? -M_PI_2 : M_PI_2; }
  }
  const double* th = thetas + (shift_mode ? 0 : bg * theta_stride);
  double U[8] = {1, 0, 0, 0, 0, 0, 1, 0};
  const uint32_t ne = fw[1];
  for (uint32_t e = 0; e < ne; ++e) {
    const uint32_t kind = fw[2 + 2 * e], par = fw[3 + 2 * e];
    double t = 0.0;
    if (par != 0xffffffffu) { t = th[par]; if (par == pshift) t += shift; }
    double M[8], R[8];
    elem_matrix(kind, t, M);
    // R = M * U
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int cidx = 0; cidx < 2; ++cidx) {
        double re = 0.0, im = 0.0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const double ar = M[(r * 2 + q) * 2], ai = M[(r * 2 + q) * 2 + 1];
          const double br = U[(q * 2 + cidx) * 2], bi = U[(q * 2 + cidx) * 2 + 1];
          re += ar * br - ai * bi;
          im += ar * bi + ai * br;
        }
        R[(r * 2 + cidx) * 2] = re; R[(r * 2 + cidx) * 2 + 1] = im;
      }
#pragma unroll
    for (int i = 0; i < 8; ++i) U[i] = R[i];
  }
  double* dst = gates + (b * slots + f) * 8;
  if (normalise) {
    // Record of circuit_pass_r3_kernel: the matrix divided by a pivot p so that one row reads (1, B) -- 12 instead of 16
    // fp64 instructions per amplitude pair:  z0 = x0 + B x1,  z1 = C x0 + D x1.  p = u00 (z = (y0, y1) / p) or, where
    // |u10| > |u00|, p = u10 with the two outputs exchanged (z = (y1, y0) / p: the kernel folds the exchange into the
    // address the pair is written to).  |p|^2 >= 1/2 for a unitary column; the product of all |p|^2 of a circuit scales
    // its probabilities back (gate_scale_kernel); a global phase does not show in |psi|^2.
    const double m0 = U[0] * U[0] + U[1] * U[1], m1 = U[4] * U[4] + U[5] * U[5];
    const bool sw = m1 > m0;
    const double pr = sw ? U[4] : U[0], pi = sw ? U[5] : U[1], pm = sw ? m1 : m0;
    const double ir = pr / pm, ii = -pi / pm;                      // 1 / p
    const double* r0 = sw ? U + 4 : U;        // the row that becomes (1, B)
    const double* r1 = sw ? U : U + 4;        // the row that becomes (C, D)
    dst[0] = r0[2] * ir - r0[3] * ii; dst[1] = r0[2] * ii + r0[3] * ir;
    dst[2] = r1[0] * ir - r1[1] * ii; dst[3] = r1[0] * ii + r1[1] * ir;
    dst[4] = r1[2] * ir - r1[3] * ii; dst[5] = r1[2] * ii + r1[3] * ir;
    dst[6] = pm;
    dst[7] = sw ? 1.0 : 0.0;
    return;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) dst[i] = U[i];
}

// scale of circuit b's probabilities under the normalised gates: product of the pivots' |p|^2, kept in slot nf of the
// circuit's gate array.  One wave per circuit: lane l multiplies gates l, l + 64, ..., then a butterfly of products (fp
// multiplication commutes exactly, so every lane ends with the same bits: deterministic).  (One thread per circuit
// walking its ~100 gates took 32 us at n = 16 -- 1.5 % of the step's circuit time -- for 577 multiplication chains.)
__global__ __launch_bounds__(64) void gate_scale_kernel(double* __restrict__ gates, int nf, int slots, int batch) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= batch) return;
  double* g = gates + (long long)b * slots * 8;
  double sc = 1.0;
  for (int f = lane; f < nf; f += 64) sc *= g[f * 8 + 6];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sc *= __shfl_xor(sc, off, 64);
  if (lane == 0) g[(long long)nf * 8] = sc;
}

// raw 2x2 matrices [count][8] -> normalised records in place (the matrix-free Stein mat-vec writes its one shared matrix raw)
__global__ void normalise_gates_kernel(double* __restrict__ gates, int count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  double* dst = gates + (long long)i * 8;
  double U[8];
  for (int e = 0; e < 8; ++e) U[e] = dst[e];
  const double m0 = U[0] * U[0] + U[1] * U[1], m1 = U[4] * U[4] + U[5] * U[5];
  const bool sw = m1 > m0;
  const double pr = sw ? U[4] : U[0], pi = sw ? U[5] : U[1], pm = sw ? m1 : m0;
  const double ir = pr / pm, ii = -pi / pm;
  const double* r0 = sw ? U + 4 : U;
  const double* r1 = sw ? U : U + 4;
  dst[0] = r0[2] * ir - r0[3] * ii; dst[1] = r0[2] * ii + r0[3] * ir;
  dst[2] = r1[0] * ir - r1[1] * ii; dst[3] = r1[0] * ii + r1[1] * ir;
  dst[4] = r1[2] * ir - r1[3] * ii; dst[5] = r1[2] * ii + r1[3] * ir;
  dst[6] = pm;
  dst[7] = sw ? 1.0 : 0.0;
}

// ------------------------------------------------------------------------------------------------
// register-level micro-ops on the 16 amplitudes a thread holds (static indices only: runtime-indexed
// arrays would go to scratch)
// ------------------------------------------------------------------------------------------------
template <int I>
__device__ __forceinline__ void op_u1(double (&ar)[16], double (&ai)[16], const double2* __restrict__ Um) {
  // the matrix is read from LDS at a wave-uniform address (broadcast), right before its use
  const double2 u00 = Um[0], u01 = Um[1], u10 = Um[2], u11 = Um[3];
  const double U[8] = {u00.x, u00.y, u01.x, u01.y, u10.x, u10.y, u11.x, u11.y};
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (j & (1 << I)) continue;
    const int j1 = j | (1 << I);
    const double x0r = ar[j], x0i = ai[j], x1r = ar[j1], x1i = ai[j1];
    ar[j] = U[0] * x0r - U[1] * x0i + U[2] * x1r - U[3] * x1i;
    ai[j] = U[0] * x0i + U[1] * x0r + U[2] * x1i + U[3] * x1r;
    ar[j1] = U[4] * x0r - U[5] * x0i + U[6] * x1r - U[7] * x1i;
    ai[j1] = U[4] * x0i + U[5] * x0r + U[6] * x1i + U[7] * x1r;
  }
}

// product of CZ gates = (-1)^{q(x)}: q(e ^ o_j) = q(e) ^ q(o_j) ^ parity(e & m_j) for the 16 slot offsets o_j
__device__ __forceinline__ void apply_sign(const uint32_t* __restrict__ Q, uint32_t e, int n, double (&ar)[16],
                                           double (&ai)[16]) {
  uint32_t acc = 0;
  for (int q = 0; q < n; ++q) {
    const uint32_t row = Q[q];
    acc ^= ((e >> q) & 1u) & (uint32_t)__popc(e & row);
  }
  const uint32_t qbits = Q[48] ^ (0u - (acc & 1u));
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const uint32_t sgn = ((qbits >> j) ^ (uint32_t)__popc(e & Q[32 + j])) & 1u;
    ar[j] = sgn ? -ar[j] : ar[j];
    ai[j] = sgn ? -ai[j] : ai[j];
  }
}

// Byte `idx` (0..15, may be a run-time but wave-uniform value) of a 16-byte table held in four
// registers: two 64-bit shifts instead of a dependent memory load.
__device__ __forceinline__ uint32_t byte16(const uint32_t (&w)[4], uint32_t idx) {
  const unsigned long long lo = (unsigned long long)w[0] | ((unsigned long long)w[1] << 32);
  const unsigned long long hi = (unsigned long long)w[2] | ((unsigned long long)w[3] << 32);
  const unsigned long long v = (idx & 8u) ? hi : lo;
  return (uint32_t)(v >> (8u * (idx & 7u))) & 0xffu;
}

// sum over bits j in [from, to) of bit_j(v) << table[j]; fully unrolled so the table stays in registers
__device__ __forceinline__ uint32_t deposit16(uint32_t v, int from, int to, const uint32_t (&table)[4]) {
  uint32_t o = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j)
    if (j >= from && j < to) o |= ((v >> j) & 1u) << ((table[j >> 2] >> (8 * (j & 3))) & 0xffu);
  return o;
}

// entry `idx` (compile-time) of a table of sixteen 16-bit values held in eight registers
__device__ __forceinline__ uint32_t half16(const uint32_t (&w)[8], int idx) {
  return (w[idx >> 1] >> (16 * (idx & 1))) & 0xffffu;
}

// xor over bits j in [0, nbits) of v of table[j]  (GF(2)-linear map of an index into a swizzled LDS slot)
__device__ __forceinline__ uint32_t xor_map16(uint32_t v, int nbits, const uint32_t (&table)[8]) {
  uint32_t o = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j)
    if (j < nbits) o ^= (0u - ((v >> j) & 1u)) & half16(table, j);
  return o;
}

// xor over bits j in [0, nbits) of v of cols[j] (32-bit columns in the plan: GF(2)-linear phys-out address)
__device__ __forceinline__ uint32_t xor_cols(uint32_t v, int nbits, const uint32_t* __restrict__ cols) {
  uint32_t o = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j)
    if (j < nbits) o ^= (0u - ((v >> j) & 1u)) & cols[j];
  return o;
}

// table entry at a run-time (wave-uniform) index
__device__ __forceinline__ uint32_t half16_dyn(const uint32_t (&w)[8], int idx) {
  uint32_t o = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) o = (j == idx) ? half16(w, j) : o;
  return o;
}

constexpr int MAX_TILE_ITERS = 16;  // tile elements per thread in the HBM <-> LDS phases (2^13 / 512)

// One pass of the circuit program over one 2^k-amplitude tile per workgroup.
// grid = (2^(n-k) tiles, circuits); block = plan threads; dynamic LDS = 2^k * 16 bytes.
// FULL = every thread moves exactly MAX_TILE_ITERS tile elements (k >= 10: threads = 2^(k-4)), which lets
// all loads of the tile stay in registers; the generic instantiation handles small tiles.
template <bool FULL, bool DEBUG>
__global__ __launch_bounds__(512) void circuit_pass_kernel(
    const uint32_t* __restrict__ plan, uint32_t pass_off, const double2* __restrict__ in,
    double2* __restrict__ out, double* __restrict__ probs, const double* __restrict__ gates,
    long long gate_stride, long long state_stride, long long total_tiles, int dbg_arg) {
  // timing-only ablation flags exist only in the DEBUG instantiation: in the production kernel every
  // `dbg` test folds away (as run-time tests they cost a scalar branch per LDS access)
  const int dbg = DEBUG ? dbg_arg : 0;
  extern __shared__ double2 tile[];
  const uint32_t* __restrict__ P = plan + pass_off;
  uint32_t H[PW_HEADER_WORDS];   // whole pass header with two wide scalar loads
#pragma unroll
  for (int i = 0; i < PW_HEADER_WORDS; ++i) H[i] = P[i];
  const uint32_t flags = H[PW_FLAGS];
  const int k = (int)H[PW_K], n = (int)H[PW_N];
  const int nstages = (int)H[PW_NSTAGES];
  const int lo_in = (int)H[PW_LO_IN], lo_out = (int)H[PW_LO_OUT];
  const uint32_t in_phys[4] = {H[PW_IN_PHYS], H[PW_IN_PHYS + 1], H[PW_IN_PHYS + 2], H[PW_IN_PHYS + 3]};
  const uint32_t in_gphys[4] = {H[PW_IN_GPHYS], H[PW_IN_GPHYS + 1], H[PW_IN_GPHYS + 2], H[PW_IN_GPHYS + 3]};
  uint32_t in_mask[8], in_gmask[8], out_mask[8], out_gmask[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    in_mask[i] = H[PW_IN_MASK + i]; in_gmask[i] = H[PW_IN_GMASK + i];
    out_mask[i] = H[PW_OUT_MASK + i]; out_gmask[i] = H[PW_OUT_GMASK + i];
  }
  const uint32_t out_phys[4] = {H[PW_OUT_PHYS], H[PW_OUT_PHYS + 1], H[PW_OUT_PHYS + 2], H[PW_OUT_PHYS + 3]};
  const uint32_t out_gphys[4] = {H[PW_OUT_GPHYS], H[PW_OUT_GPHYS + 1], H[PW_OUT_GPHYS + 2], H[PW_OUT_GPHYS + 3]};
  const uint32_t t = threadIdx.x, T = blockDim.x;
  const int tau = 31 - __clz((int)T);
  const int kt = tau < k ? tau : k;  // index bits supplied by the thread id
  const uint32_t ksize = 1u << k;
  const int niter = 1 << (k - kt);    // <= MAX_TILE_ITERS (checked by the planner: threads >= 2^(k-4))
  double2* __restrict__ mats = tile + ksize;   // [nstages][4 register bits][4 double2] fused matrices of circuit b
  const int gbits = n - k;

  // A workgroup walks over tiles T = blockIdx.x, blockIdx.x + gridDim.x, ... of the flattened (circuit, tile)
  // space.  With gridDim.x == total_tiles every workgroup handles one tile; with a PERSISTENT grid (as many
  // workgroups as are co-resident) the header is fetched once per workgroup and a tile's write-back drains
  // while the next tile's loads are already in flight.
  for (long long Tcur = blockIdx.x; Tcur < total_tiles; Tcur += gridDim.x) {
  const uint32_t g = (uint32_t)(Tcur & ((1ll << gbits) - 1));
  const long long b = Tcur >> gbits;

  // ---- this circuit's fused matrices for every stage of the pass -> LDS (one 16-byte piece per thread;
  // they were written by build_gates_kernel on other CUs, i.e. they come from far memory: fetch them once,
  // under the tile loads, instead of stalling every stage on them)
  for (uint32_t piece = t; piece < (uint32_t)nstages * 16u && !(dbg & 32); piece += T) {
    const uint32_t sm = piece >> 2;                               // (stage, register bit)
    const uint32_t w = P[PW_MATS + (sm >> 1)];
    const uint32_t f = (sm & 1u) ? (w >> 16) : (w & 0xffffu);
    if (f != 0xffffu)
      mats[piece] = reinterpret_cast<const double2*>(gates + b * gate_stride + (size_t)f * 8)[piece & 3u];
  }

  // ---- tile in: HBM (pass-specific bit order) -> LDS, or |0...0> ---------------------------------
  if (flags & PASS_INIT) {
    for (uint32_t u = t; u < ksize; u += T) tile[u] = make_double2((u == 0 && g == 0) ? 1.0 : 0.0, 0.0);
  } else if (t < ksize && !(dbg & 4)) {
    const uint32_t gin = deposit16(g, 0, n - k, in_gphys);
    const double2* __restrict__ src = in + b * state_stride + gin;
    const uint32_t thr = (t & ((1u << lo_in) - 1u)) | deposit16(t, lo_in, kt, in_phys);
    uint32_t ipos[4], imask[4];       // phys-in position / LDS slot mask of the index bits supplied by the iteration count
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      ipos[m] = (kt + m < k) ? byte16(in_phys, (uint32_t)(kt + m)) : 0u;
      imask[m] = (kt + m < k) ? half16_dyn(in_mask, kt + m) : 0u;
    }
    // swizzled LDS slot of element u = slot_t ^ (iteration part): the CNOTs at the head of the pass are folded in
    const uint32_t slot_t = xor_map16(t, kt, in_mask) ^ xor_map16(g, n - k, in_gmask);
    if (FULL) {
      double2 v[MAX_TILE_ITERS];      // every load of the tile is in flight before the first LDS write
#pragma unroll
      for (int i = 0; i < MAX_TILE_ITERS; ++i) {
        const uint32_t itp = ((i & 1) ? 1u << ipos[0] : 0u) | ((i & 2) ? 1u << ipos[1] : 0u) |
                             ((i & 4) ? 1u << ipos[2] : 0u) | ((i & 8) ? 1u << ipos[3] : 0u);
        v[i] = src[thr | itp];
      }
#pragma unroll
      for (int i = 0; i < MAX_TILE_ITERS; ++i) {
        const uint32_t its = ((i & 1) ? imask[0] : 0u) ^ ((i & 2) ? imask[1] : 0u) ^ ((i & 4) ? imask[2] : 0u) ^ ((i & 8) ? imask[3] : 0u);
        tile[slot_t ^ its] = v[i];
      }
    } else {
      for (int i = 0; i < niter; ++i) {
        const uint32_t itp = ((i & 1) ? 1u << ipos[0] : 0u) | ((i & 2) ? 1u << ipos[1] : 0u) |
                             ((i & 4) ? 1u << ipos[2] : 0u) | ((i & 8) ? 1u << ipos[3] : 0u);
        const uint32_t its = ((i & 1) ? imask[0] : 0u) ^ ((i & 2) ? imask[1] : 0u) ^ ((i & 4) ? imask[2] : 0u) ^ ((i & 8) ? imask[3] : 0u);
        tile[slot_t ^ its] = src[thr | itp];
      }
    }
  }
  __syncthreads();

  // ---- stages: 2^r amplitudes per thread in registers, one LDS round trip each -----------------------
  // Fixed form (plan.hpp): read with the phase-0 CNOT permutation folded into the address, optional sign,
  // one fused U per register wire (matrices in LDS), optional sign on the permuted index, write with the
  // phase-3 CNOT permutation folded into the address.  All per-stage index arithmetic was done by the
  // planner: a thread fetches its (swizzled base | base << 16) word from the stage's table -- one stage ahead,
  // so the L2 latency hides under the current stage -- and the 32 slot offsets come as whole scalar words.
  // FULL implies r == 4 (k >= 10): no per-slot guards.
  const uint32_t* __restrict__ S = P + PW_STAGES;
  uint32_t base_word = 0;
  {
    const uint32_t hdr0 = nstages > 0 ? S[0] : 0u;
    const uint32_t npay0 = (((hdr0 >> 8) & STAGE_SIGN_PRE) ? SIGNQ_WORDS : 0) + (((hdr0 >> 8) & STAGE_SIGN_POST) ? SIGNQ_WORDS : 0);
    if (nstages > 0 && t < (1u << (k - (int)(hdr0 & 0xffu)))) base_word = S[STAGE_HDR_WORDS + npay0 + t];
  }
  for (int s = 0; s < nstages && !(dbg & 64); ++s) {
    uint32_t G[STAGE_HDR_WORDS];     // stage header with three wide scalar loads
#pragma unroll
    for (int i = 0; i < STAGE_HDR_WORDS; ++i) G[i] = S[i];
    const uint32_t hdr = G[0];
    const int r = FULL ? 4 : (int)(hdr & 0xffu);
    const uint32_t sflags = (hdr >> 8) & 0xffu;
    const uint32_t nwords = hdr >> 16;
    const uint32_t my_word = base_word;
    if (s + 1 < nstages) {           // prefetch the next stage's table entry
      const uint32_t* __restrict__ Sn = S + nwords;
      const uint32_t hdrn = Sn[0];
      const uint32_t npayn = (((hdrn >> 8) & STAGE_SIGN_PRE) ? SIGNQ_WORDS : 0) + (((hdrn >> 8) & STAGE_SIGN_POST) ? SIGNQ_WORDS : 0);
      if (t < (1u << (k - (int)(hdrn & 0xffu)))) base_word = Sn[STAGE_HDR_WORDS + npayn + t];
    }
    // (two halves around an optional workgroup barrier: a stage whose read map crosses thread groups -- plan.hpp:
    // STAGE_CROSS_READ -- must not write before every thread has read)
    const bool stage_active = t < (1u << (k - r));
    const int nreg = 1 << r;
    uint32_t wbase = 0;
    double ar[16], ai[16];
    if (stage_active) {
      const uint32_t rho = G[1];
      const uint32_t fi[4] = {G[6] & 0xffffu, G[6] >> 16, G[7] & 0xffffu, G[7] >> 16};
      const double2* __restrict__ Us = mats + s * 16;
      const uint32_t pb = my_word & 0xffffu;
      const uint32_t e = (my_word >> 16) | (g << k);  // extended index: LDS bits then workgroup bits
      uint32_t lflip = 0, sflip = 0, e2 = e;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t bpre = G[8 + i], bpost = G[12 + i];
        if (bpre | bpost) {
          const uint32_t rp = (rho >> (8 * i)) & 0xffu;
          const uint32_t sri = lds_swizzle(1u << rp);
          if (bpre) lflip ^= (__popc(e & bpre) & 1) ? sri : 0u;
          if (bpost) {
            const uint32_t bit = (uint32_t)__popc(e & bpost) & 1u;
            sflip ^= bit ? sri : 0u;
            e2 |= bit << rp;
          }
        }
      }
#pragma unroll
      for (int p2 = 0; p2 < 16; ++p2) {        // phase-0 CNOTs whose target is a thread-held position (plan.hpp: BpreF)
        const uint32_t m = G[48 + p2];
        if (m) lflip ^= (__popc(e & m) & 1) ? lds_swizzle(1u << p2) : 0u;
      }
      const uint32_t rbase = pb ^ lflip;
      wbase = pb ^ sflip;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if ((FULL || j < nreg) && !(dbg & 2)) { const double2 x = tile[rbase ^ G[16 + j]]; ar[j] = x.x; ai[j] = x.y; }
        else { ar[j] = (double)(rbase + j); ai[j] = 0.0; }
      }
      const uint32_t* __restrict__ Q = S + STAGE_HDR_WORDS;
      if ((sflags & STAGE_SIGN_PRE) && !(dbg & 128)) { apply_sign(Q, e, n, ar, ai); Q += SIGNQ_WORDS; }
      if (!(dbg & 1)) {
        if (fi[0] != 0xffffu) op_u1<0>(ar, ai, Us);
        if (fi[1] != 0xffffu) op_u1<1>(ar, ai, Us + 4);
        if (fi[2] != 0xffffu) op_u1<2>(ar, ai, Us + 8);
        if (fi[3] != 0xffffu) op_u1<3>(ar, ai, Us + 12);
      }
      if ((sflags & STAGE_SIGN_POST) && !(dbg & 128)) apply_sign(Q, e2, n, ar, ai);
    }
    if (sflags & STAGE_CROSS_READ) __syncthreads();
    if (stage_active) {
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if ((FULL || j < nreg) && !(dbg & 2)) tile[wbase ^ G[32 + j]] = make_double2(ar[j], ai[j]);
      if (dbg & 2) {   // timing-only build of the stage without LDS traffic: keep the values alive
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc += ar[j] + ai[j];
        if (acc == 1.2345e300) tile[0] = make_double2(acc, acc);
      }
    }
    if (!(dbg & 16)) __syncthreads();
    S += nwords;
  }

  // ---- tile out: LDS -> HBM in the next pass's bit order, or |psi|^2 in canonical order ---------------
  if (t < ksize && !(dbg & 8)) {
    // phys-out address: GF(2)-linear in the out-enumeration index and the tile index (plan.hpp: PW_OUT_COL)
    const uint32_t gout = xor_cols(g, n - k, P + PW_OUT_GCOL);
    const uint32_t thr_l = xor_map16(t, kt, out_mask) ^ xor_map16(g, n - k, out_gmask);   // tail CNOTs folded in
    const uint32_t thr_p = xor_cols(t, kt, P + PW_OUT_COL) ^ gout;
    uint32_t lpos[4], pcol[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      lpos[m] = (kt + m < k) ? half16_dyn(out_mask, kt + m) : 0u;
      pcol[m] = (kt + m < k) ? P[PW_OUT_COL + kt + m] : 0u;
    }
    const bool fin = flags & PASS_FINAL;
    double2* __restrict__ dst = out + b * state_stride;
    double* __restrict__ pdst = probs + (b << n);
    auto move_out = [&](int i) {
      const uint32_t it_l = ((i & 1) ? lpos[0] : 0u) ^ ((i & 2) ? lpos[1] : 0u) ^ ((i & 4) ? lpos[2] : 0u) ^ ((i & 8) ? lpos[3] : 0u);
      const uint32_t it_p = ((i & 1) ? pcol[0] : 0u) ^ ((i & 2) ? pcol[1] : 0u) ^ ((i & 4) ? pcol[2] : 0u) ^ ((i & 8) ? pcol[3] : 0u);
      const double2 v = tile[thr_l ^ it_l];
      if (fin) pdst[thr_p ^ it_p] = v.x * v.x + v.y * v.y;
      else dst[thr_p ^ it_p] = v;
    };
    if (FULL) {
#pragma unroll
      for (int i = 0; i < MAX_TILE_ITERS; ++i) move_out(i);
    } else {
      for (int i = 0; i < niter; ++i) move_out(i);
    }
  }
  if (Tcur + gridDim.x < total_tiles) __syncthreads();   // the tile and the matrices are overwritten by the next tile
  }  // tile loop
}

// ------------------------------------------------------------------------------------------------
// Fast pass kernel (tiles of 2^10 .. 2^13 amplitudes, 16 per thread).  Same plan, same arithmetic model as
// circuit_pass_kernel, restructured around what the profile showed:
//   * PERSISTENT workgroups walk over the (circuit, tile) space; while a tile is in its stage loop the NEXT
//     tile's amplitudes and matrices are already in flight into registers and the previous tile's stores drain,
//     so the HBM phases overlap the FMA/LDS phases inside one workgroup (and workgroup dispatch, header and
//     index-map set-up are paid once per workgroup, not once per tile);
//   * every index-dependent quantity of a stage (LDS slots with the CNOT flips folded in, CZ sign bits) comes
//     from planner tables RW / SG (plan.hpp, build_fast_tables): one coalesced 4-byte load per thread per
//     stage, fetched one stage ahead; the 16 slot addresses follow from 4 basis offsets by one xor each;
//   * the 2x2 complex gate updates its amplitude pair IN PLACE (12 partial-sum ops into 4 temporaries, then 4
//     final FMAs that each overwrite the operand they read last): the compiler's version needed 32 register
//     moves per gate to merge the conditional gate back into the amplitude registers.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double uniform_to_sgpr(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

template <int I>
__device__ __forceinline__ void op_u1_inplace_s(double (&ar)[16], double (&ai)[16], const double (&U)[8]) {
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (j & (1 << I)) continue;
    gate_pair_inplace_s(ar[j], ai[j], ar[j | (1 << I)], ai[j | (1 << I)], U);
  }
}

#ifndef BORNVI_U_SGPR
#define BORNVI_U_SGPR 0
#endif

template <int I>
__device__ __forceinline__ void op_u1_inplace(double (&ar)[16], double (&ai)[16], const double (&U)[8]) {
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (j & (1 << I)) continue;
    gate_pair_inplace(ar[j], ai[j], ar[j | (1 << I)], ai[j | (1 << I)], U);
  }
}

// flips the sign of slot j where bit j of `m` is set (xor on the sign bit of both components)
__device__ __forceinline__ void apply_sign_bits(uint32_t m, double (&ar)[16], double (&ai)[16]) {
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int sw = (int)((m << (31 - j)) & 0x80000000u);
    ar[j] = __hiloint2double(__double2hiint(ar[j]) ^ sw, __double2loint(ar[j]));
    ai[j] = __hiloint2double(__double2hiint(ai[j]) ^ sw, __double2loint(ai[j]));
  }
}


// The LAST gate of an LDS -> LDS stage with the write-back folded in: as soon as a pair has gone through the gate (and
// the post sign) its two amplitudes are final and go to LDS, while the next pair is still in the FMAs.  All waves of
// a large-tile workgroup reach the end of a stage together; 16 ds_write_b128 per wave issued in one burst by all of
// them kept the LDS write path (~80 B/clk/CU) busy for ~1600 cycles per stage with the VALU idle.  Same operations on
// the same values: bit-identical results.
#ifndef BORNVI_SKIP_ZERO_TILES
#define BORNVI_SKIP_ZERO_TILES 1   // 0: run the first pass's all-zero tiles through the stages like any other (A/B)
#endif
#ifndef BORNVI_GATE_PRIO
#define BORNVI_GATE_PRIO 0       // experiment: s_setprio level while a wave runs its gates (0 = off)
#endif
#if BORNVI_GATE_PRIO
#define BORNVI_PRIO_OFF() __builtin_amdgcn_s_setprio(0)
#else
#define BORNVI_PRIO_OFF() do { } while (0)
#endif
#ifndef BORNVI_EARLY_WRITE
#define BORNVI_EARLY_WRITE 1     // large-tile instantiation only (LT below); 0: round-1 stage body everywhere (A/B)
#endif
#ifndef BORNVI_EARLY_STORE
#define BORNVI_EARLY_STORE 1     // the same for a direct last stage's HBM stores
#endif
#ifndef BORNVI_U_PREFETCH
#define BORNVI_U_PREFETCH 1      // with BORNVI_EARLY_WRITE: two matrix register sets, matrix reads ahead of the amplitude reads
#endif
template <int I, bool POST>
__device__ __forceinline__ void op_u1_last_and_write(double (&ar)[16], double (&ai)[16], const double (&U)[8], uint32_t post_bits,
                                                     double2* __restrict__ tile, uint32_t wa0, const uint32_t (&G)[10]) {
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (j & (1 << I)) continue;
    const int j1 = j | (1 << I);
    gate_pair_inplace(ar[j], ai[j], ar[j1], ai[j1], U);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int jj = q ? j1 : j;
      double xr = ar[jj], xi = ai[jj];
      if (POST) {
        const int sw = (int)((post_bits << (31 - jj)) & 0x80000000u);
        xr = __hiloint2double(__double2hiint(xr) ^ sw, __double2loint(xr));
        xi = __hiloint2double(__double2hiint(xi) ^ sw, __double2loint(xi));
      }
      const uint32_t wa = wa0 ^ (((jj & 1) ? G[FS_WB] : 0u) ^ ((jj & 2) ? G[FS_WB + 1] : 0u) ^ ((jj & 4) ? G[FS_WB + 2] : 0u) ^ ((jj & 8) ? G[FS_WB + 3] : 0u));
      *reinterpret_cast<double2*>(reinterpret_cast<char*>(tile) + wa) = make_double2(xr, xi);
    }
  }
}


// The same for a direct last stage (IO == 2): the pair's two results go straight to HBM behind its gate -- the 16
// stores of the tile are spread over the last gate instead of one burst per wave (a 1-KiB store takes ~100 cycles of the
// CU's memory pipe).  Still exactly 16 stores per wave and tile, all behind the trip's loads.
#ifndef BORNVI_FIN_HOIST
#define BORNVI_FIN_HOIST 1      // one test of `fin` per last gate (two copies of the gate) instead of one per store: -1.3 %
#endif
template <int I, bool POST, int FIN = -1>
__device__ __forceinline__ void op_u1_last_and_store(double (&ar)[16], double (&ai)[16], const double (&U)[8], uint32_t post_bits,
                                                     uint32_t ha0, const uint32_t (&hbm_basis)[4], void* hbm_base, bool fin) {
  if (BORNVI_FIN_HOIST && FIN < 0) {
    if (fin) op_u1_last_and_store<I, POST, 1>(ar, ai, U, post_bits, ha0, hbm_basis, hbm_base, true);
    else op_u1_last_and_store<I, POST, 0>(ar, ai, U, post_bits, ha0, hbm_basis, hbm_base, false);
    return;
  }
  if (FIN >= 0) fin = FIN != 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (j & (1 << I)) continue;
    const int j1 = j | (1 << I);
    gate_pair_inplace(ar[j], ai[j], ar[j1], ai[j1], U);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int jj = q ? j1 : j;
      double xr = ar[jj], xi = ai[jj];
      if (POST) {
        const int sw = (int)((post_bits << (31 - jj)) & 0x80000000u);
        xr = __hiloint2double(__double2hiint(xr) ^ sw, __double2loint(xr));
        xi = __hiloint2double(__double2hiint(xi) ^ sw, __double2loint(xi));
      }
      const uint32_t ha = ha0 ^ (((jj & 1) ? hbm_basis[0] : 0u) ^ ((jj & 2) ? hbm_basis[1] : 0u) ^ ((jj & 4) ? hbm_basis[2] : 0u) ^ ((jj & 8) ? hbm_basis[3] : 0u));
      if (fin) async_store8(ha, xr * xr + xi * xi, hbm_base);
      else async_store16(ha, (d2_t){xr, xi}, hbm_base);
    }
  }
}

// One stage on the 16 amplitudes a thread owns, specialised on the number of fused gates (the planner puts
// them on register bits 0 .. NG-1) and on the two CZ sign products: straight-line code, no merges of the
// 64 amplitude registers (conditional gates cost ~100 register copies per stage in the generic kernel).
// IO: 0 = LDS -> LDS;  1 = the amplitudes are the prefetched registers `v` (first stage of a pass: no tile
// fill, no LDS read);  2 = the results go straight to HBM (last stage: no LDS write, no tile drain): 16 stores
// at  hbm_off ^ (xor of hbm_basis over the bits of the slot number), |amp|^2 as 8 bytes when `fin`.
// LT: large-tile instantiation (one workgroup of >= 4 waves per CU, all waves in step between barriers): the stage
// body folds the write-back into the last gate and keeps the next gate's matrix in flight.  Measured at n = 20, L = 8:
// -2.4 % (early write-back) and -5.2 % (both) on top of the split prefetch; at 2^11 tiles (four independent
// workgroups per CU) neutral within noise, so those keep the round-1 body.  Same operations: bit-identical rows.
template <int NG, bool PRE, bool POST, int IO, bool DEBUG, bool LT>
__device__ __forceinline__ void stage_body(double2* __restrict__ tile, const double2* __restrict__ Us, uint32_t my_rw,
                                           uint32_t my_sg, const uint32_t (&G)[10], int dbg, d2_t (&v)[16],
                                           uint32_t hbm_off, const uint32_t (&hbm_basis)[4], void* hbm_base, bool fin,
                                           bool cross) {
  double ar[16], ai[16];
#if BORNVI_U_PREFETCH
  // (first: LDS returns in order, so the first gate can start as soon as ITS amplitudes have arrived behind the matrix)
  double Ua[8], Ub[8];
  constexpr bool EARLY = BORNVI_EARLY_WRITE && LT && !DEBUG && NG > 0 && (IO != 2 || BORNVI_EARLY_STORE);
  if (EARLY) {
    load_u(Us, Ua);
    if (NG > 1) load_u(Us + 4, Ub);
  }
#endif
  if (IO == 1) {
#pragma unroll
    for (int j = 0; j < 16; ++j) { ar[j] = v[j].x; ai[j] = v[j].y; }
  } else {
    // byte addresses of the 16 slots: GF(2)-linear in the slot number -- one per-thread base, the 16 offsets are
    // wave-uniform (scalar registers), one v_xor per access and no address kept live across the gates
    const uint32_t ra0 = (my_rw & 0xffffu) << 4;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const uint32_t ra = ra0 ^ (((j & 1) ? G[FS_RB] : 0u) ^ ((j & 2) ? G[FS_RB + 1] : 0u) ^ ((j & 4) ? G[FS_RB + 2] : 0u) ^ ((j & 8) ? G[FS_RB + 3] : 0u));
      if (!(DEBUG && (dbg & 2))) {
        const double2 x = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(tile) + ra);
        ar[j] = x.x; ai[j] = x.y;
      } else { ar[j] = (double)(ra + j); ai[j] = 0.0; }
    }
  }
  if (IO != 1 && cross) {
    // the read map took amplitudes from other threads' groups (plan.hpp: STAGE_CROSS_READ): nobody writes before
    // everybody has its 16 amplitudes (wave-uniform flag, same for the whole workgroup)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (PRE) apply_sign_bits(my_sg & 0xffffu, ar, ai);
#if BORNVI_GATE_PRIO
  if (LT && NG > 0) __builtin_amdgcn_s_setprio(BORNVI_GATE_PRIO);   // experiment: the wave inside its gates wins the issue arbitration
#endif
  if (BORNVI_EARLY_WRITE && LT && !DEBUG && NG > 0 && (IO != 2 || BORNVI_EARLY_STORE)) {
    constexpr int LAST = NG > 0 ? NG - 1 : 0;
    uint32_t wa0 = (IO == 2) ? hbm_off : (my_rw >> 16) << 4;
    // the last gate with its write-back: to LDS, or (direct last stage) to HBM
#define BORNVI_LAST_GATE(U_)                                                                                   \
    do {                                                                                                       \
      asm volatile("" : "+v"(wa0));   /* the write base is formed here, before the last gate */                \
      if (IO == 2) op_u1_last_and_store<LAST, POST>(ar, ai, U_, my_sg >> 16, wa0, hbm_basis, hbm_base, fin);   \
      else op_u1_last_and_write<LAST, POST>(ar, ai, U_, my_sg >> 16, tile, wa0, G);                            \
      BORNVI_PRIO_OFF();                                                                                        \
    } while (0)
#if BORNVI_U_PREFETCH
    // two matrix register sets: the next gate's four broadcast LDS reads are in flight under the current gate's FMAs
    // (the early write-back freed the registers for the second set)
    if (NG == 1) { BORNVI_LAST_GATE(Ua); return; }
    op_u1_inplace<0>(ar, ai, Ua);
    if (NG > 2) load_u(Us + 8, Ua);
    if (NG == 2) { BORNVI_LAST_GATE(Ub); return; }
    op_u1_inplace<1>(ar, ai, Ub);
    if (NG > 3) load_u(Us + 12, Ub);
    if (NG == 3) { BORNVI_LAST_GATE(Ua); return; }
    op_u1_inplace<2>(ar, ai, Ua);
    BORNVI_LAST_GATE(Ub);
    return;
#else
    double U[8];
    if (NG > 1) { load_u(Us, U); op_u1_inplace<0>(ar, ai, U); }
    if (NG > 2) { load_u(Us + 4, U); op_u1_inplace<1>(ar, ai, U); }
    if (NG > 3) { load_u(Us + 8, U); op_u1_inplace<2>(ar, ai, U); }
    load_u(Us + 4 * LAST, U);
    BORNVI_LAST_GATE(U);
    return;
#endif
#undef BORNVI_LAST_GATE
  }
  if (!(DEBUG && (dbg & 1))) {
    double U[8];
#if BORNVI_U_SGPR
    // the wave-uniform matrix goes from LDS through a VGPR staging set into scalar registers; the next gate's LDS reads
    // are issued into the freed staging set before this gate's FMAs (their latency hides under them)
    double S_[8];
    if (NG > 0) load_u(Us, U);
#define BORNVI_GATE_S(I_)                                                            \
    if (NG > I_) {                                                                   \
      _Pragma("unroll") for (int e_ = 0; e_ < 8; ++e_) S_[e_] = uniform_to_sgpr(U[e_]); \
      if (NG > I_ + 1) load_u(Us + 4 * (I_ + 1), U);                                 \
      op_u1_inplace_s<I_>(ar, ai, S_);                                               \
    }
    BORNVI_GATE_S(0) BORNVI_GATE_S(1) BORNVI_GATE_S(2) BORNVI_GATE_S(3)
#undef BORNVI_GATE_S
#else
    if (NG > 0) { load_u(Us, U); op_u1_inplace<0>(ar, ai, U); }
    if (NG > 1) { load_u(Us + 4, U); op_u1_inplace<1>(ar, ai, U); }
    if (NG > 2) { load_u(Us + 8, U); op_u1_inplace<2>(ar, ai, U); }
    if (NG > 3) { load_u(Us + 12, U); op_u1_inplace<3>(ar, ai, U); }
#endif
  }
  if (POST) apply_sign_bits(my_sg >> 16, ar, ai);
  if (IO == 2) {
    uint32_t ha0 = hbm_off;
    asm volatile("" : "+v"(ha0));     // addresses are formed here, after the gates
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (DEBUG && (dbg & 8)) continue;
      const uint32_t ha = ha0 ^ (((j & 1) ? hbm_basis[0] : 0u) ^ ((j & 2) ? hbm_basis[1] : 0u) ^ ((j & 4) ? hbm_basis[2] : 0u) ^ ((j & 8) ? hbm_basis[3] : 0u));
      if (fin) async_store8(ha, ar[j] * ar[j] + ai[j] * ai[j], hbm_base);
      else async_store16(ha, (d2_t){ar[j], ai[j]}, hbm_base);
    }
  } else {
    uint32_t wa0 = (my_rw >> 16) << 4;
    asm volatile("" : "+v"(wa0));     // addresses are formed here, after the gates
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const uint32_t wa = wa0 ^ (((j & 1) ? G[FS_WB] : 0u) ^ ((j & 2) ? G[FS_WB + 1] : 0u) ^ ((j & 4) ? G[FS_WB + 2] : 0u) ^ ((j & 8) ? G[FS_WB + 3] : 0u));
      if (!(DEBUG && (dbg & 2))) *reinterpret_cast<double2*>(reinterpret_cast<char*>(tile) + wa) = make_double2(ar[j], ai[j]);
    }
  }
  if (DEBUG && (dbg & 2)) {
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc += ar[j] + ai[j];
    if (acc == 1.2345e300) tile[0] = make_double2(acc, acc);
  }
}

// all (gate count, pre sign, post sign) kinds of one IO mode behind one scalar switch
template <int IO, bool DEBUG, bool LT>
__device__ __forceinline__ void stage_dispatch(uint32_t kind, double2* __restrict__ tile, const double2* __restrict__ Us,
                                               uint32_t my_rw, uint32_t my_sg, const uint32_t (&G)[10], int dbg,
                                               d2_t (&v)[16], uint32_t hbm_off, const uint32_t (&hbm_basis)[4],
                                               void* hbm_base, bool fin, bool cross) {
#define BORNVI_STAGE(NG, PRE, POST) \
  case (NG) | ((PRE) << 3) | ((POST) << 4): \
    stage_body<NG, PRE, POST, IO, DEBUG, LT>(tile, Us, my_rw, my_sg, G, dbg, v, hbm_off, hbm_basis, hbm_base, fin, cross); break;
#define BORNVI_STAGE_NG(PRE, POST) \
  BORNVI_STAGE(0, PRE, POST) BORNVI_STAGE(1, PRE, POST) BORNVI_STAGE(2, PRE, POST) BORNVI_STAGE(3, PRE, POST) BORNVI_STAGE(4, PRE, POST)
#if BORNVI_TIMING_NO_GATES   /* experiment (wrong results): the stages' LDS round trips and signs without the gate arithmetic */
  kind &= ~7u;
#endif
  switch (kind) {
    BORNVI_STAGE_NG(0, 0)
    BORNVI_STAGE_NG(1, 0)
    BORNVI_STAGE_NG(0, 1)
    BORNVI_STAGE_NG(1, 1)
    default: break;   // unreachable: build_fast_tables admits only plans whose kinds pass fast_stage_kind_supported
  }
  static_assert(fast_stage_kind_supported(4u | 8u | 16u) && !fast_stage_kind_supported(5u) && !fast_stage_kind_supported(32u),
                "the switch above and plan.hpp's fast_stage_kind_supported must list the same kinds");
#undef BORNVI_STAGE_NG
#undef BORNVI_STAGE
}

// In-kernel phase stamps (diagnostic build only, -DBORNVI_STAMPS=1, tools/probes/stamp_probe.py): wave 0 of every
// workgroup adds up, per phase of a tile trip, the shader cycles (s_memtime) it spent there; totals go to a buffer that
// nothing else reads.  In the production build every BORNVI_STAMP expands to nothing.
#ifndef BORNVI_STAMPS
#define BORNVI_STAMPS 0
#endif
#ifndef BORNVI_SPLIT_PREFETCH
#define BORNVI_SPLIT_PREFETCH 1      // 0: never use the SPLIT instantiation (A/B against the round-1 form)
#endif
#if BORNVI_STAMPS
__device__ unsigned long long g_stamp_totals[16];
#define BORNVI_STAMP(PH_)                                                        \
  do {                                                                           \
    const unsigned long long now_ = __builtin_readcyclecounter();                \
    stamp_acc[PH_] += now_ - stamp_last;                                         \
    stamp_last = now_;                                                           \
  } while (0)
#else
#define BORNVI_STAMP(PH_) do { } while (0)
#endif

template <bool DEBUG, bool SPLIT>
__global__ __launch_bounds__(512) void circuit_pass_fast_kernel(
    const uint32_t* __restrict__ plan, uint32_t pass_off, const uint32_t* __restrict__ fast, uint32_t fast_off,
    const double2* __restrict__ in, double2* __restrict__ out, double* __restrict__ probs,
    const double* __restrict__ gates, long long gate_stride, long long state_stride, long long total_tiles,
    uint32_t lds_tab_off /* double2 units */, uint32_t lds_mats2_off /* double2 units */, int direct_mask, int dbg_arg,
    PrefixShare share) {
  const int dbg = DEBUG ? dbg_arg : 0;
  extern __shared__ double2 tile[];
  const uint32_t* __restrict__ P = plan + pass_off;
  const uint32_t* __restrict__ F = fast + fast_off;
  uint32_t H[PW_HEADER_WORDS];
#pragma unroll
  for (int i = 0; i < PW_HEADER_WORDS; ++i) H[i] = P[i];
  const uint32_t flags = H[PW_FLAGS];
  const int k = (int)H[PW_K], n = (int)H[PW_N];
  const int lo_in = (int)H[PW_LO_IN], lo_out = (int)H[PW_LO_OUT];
  const uint32_t in_phys[4] = {H[PW_IN_PHYS], H[PW_IN_PHYS + 1], H[PW_IN_PHYS + 2], H[PW_IN_PHYS + 3]};
  const uint32_t in_gphys[4] = {H[PW_IN_GPHYS], H[PW_IN_GPHYS + 1], H[PW_IN_GPHYS + 2], H[PW_IN_GPHYS + 3]};
  const uint32_t out_phys[4] = {H[PW_OUT_PHYS], H[PW_OUT_PHYS + 1], H[PW_OUT_PHYS + 2], H[PW_OUT_PHYS + 3]};
  const uint32_t out_gphys[4] = {H[PW_OUT_GPHYS], H[PW_OUT_GPHYS + 1], H[PW_OUT_GPHYS + 2], H[PW_OUT_GPHYS + 3]};
  uint32_t in_mask[8], in_gmask[8], out_mask[8], out_gmask[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    in_mask[i] = H[PW_IN_MASK + i]; in_gmask[i] = H[PW_IN_GMASK + i];
    out_mask[i] = H[PW_OUT_MASK + i]; out_gmask[i] = H[PW_OUT_GMASK + i];
  }
  const uint32_t t = threadIdx.x, T = blockDim.x;   // T == 2^(k-4): 16 tile elements per thread
  const int kt = k - 4;
  const uint32_t ksize = 1u << k;
  const int gbits = n - k;
  const int nstages = (int)F[FH_NSTAGES];
  const uint32_t rw_base = F[FH_RW_BASE], sg_base = F[FH_SG_BASE];
  const uint32_t sign_pre = F[FH_SIGN_PRE], sign_post = F[FH_SIGN_POST];
  const bool any_sign = (sign_pre | sign_post) != 0;
  const uint32_t stage_words = 1u << (n - 4);     // table words per stage
  double2* __restrict__ mats = tile + ksize;      // [nstages][4 register bits][4 double2]
  uint32_t* __restrict__ tab_rw = reinterpret_cast<uint32_t*>(tile + lds_tab_off);   // [nstages][T] of this tile row
  uint32_t* __restrict__ tab_sg = tab_rw + (uint32_t)nstages * T;    // rows of the sign stages only, in stage order
  const bool init = flags & PASS_INIT, fin = flags & PASS_FINAL;

  // ---- the tile <-> HBM maps: the thread parts (thr_in, slot_in, slot_out, thr_out below) are recomputed from the
  // thread id where they are used -- a dozen VALU ops per tile; kept in registers across the tile loop they were
  // spilled, and every scratch reload drains vmcnt, i.e. waits for the prefetch in flight ----
  const int out_shift = fin ? 3 : 4;   // bytes per element written
  uint32_t ipos[4], imask[4], lpos[4], pcol[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    ipos[m] = byte16(in_phys, (uint32_t)(kt + m));
    imask[m] = half16_dyn(in_mask, kt + m);
    lpos[m] = half16_dyn(out_mask, kt + m);
    pcol[m] = P[PW_OUT_COL + kt + m] << out_shift;
  }
  // the 16-byte piece of the pass's matrices this thread stages per tile (the planner admits at most T pieces);
  // a slot without a gate copies a piece of fused gate 0: never read, and no per-thread predicate to keep.
  // Per-thread words that must survive the tile loop live in LDS (tw_*): kept in registers they were spilled, and
  // a scratch reload drains vmcnt -- it waits for the prefetch in flight and for the last tile's store acks.
  const uint32_t npieces = (uint32_t)nstages * 16u;
  const uint32_t sign_any = sign_pre | sign_post;
  uint32_t* __restrict__ tw_mat = tab_sg + (uint32_t)__popc(sign_any) * T;   // byte offset of the piece in the gate array
  uint32_t* __restrict__ tw_slots = tw_mat + T;  // thread part of the tile-fill slot | of the tile-drain slot << 16
  uint32_t* __restrict__ tw_in = tw_slots + T;   // byte offset of the thread's first load: phys-in thread part, or (direct
                                                 // first stage) of slot 0 of its first-stage group for tile row g_pref
  uint32_t* __restrict__ tw_out = tw_in + T;     // same on the way out (thread part of phys-out | last-stage group, row g_tab)
  {
    uint32_t mo = 0;
    if (t < npieces) {
      const uint32_t sm = t >> 2;
      const uint32_t w = P[PW_MATS + (sm >> 1)];
      const uint32_t f = (sm & 1u) ? (w >> 16) : (w & 0xffffu);
      mo = ((f != 0xffffu ? f : 0u) * 4u + (t & 3u)) << 4;
    }
    tw_mat[t] = mo;
  }
  // ---- direct HBM <-> register stages (plan.hpp: FH_IN_TAB / FH_OUT_TAB) ----
  const uint32_t in_tab = F[FH_IN_TAB], out_tab = F[FH_OUT_TAB];
  // (direct_mask: bit 0 / 1 allow the direct first / last stage -- an A/B switch, bornvi_set_option "direct_stages")
  const bool direct_in = in_tab != 0 && !init && nstages > 0 && (direct_mask & 1);     // first stage: amplitudes straight from HBM
  const bool direct_out = out_tab != 0 && nstages > 1 && (direct_mask & 2);   // last stage (not also the first): results straight to HBM
  const uint32_t in_basis[4] = {F[FH_IN_BASIS], F[FH_IN_BASIS + 1], F[FH_IN_BASIS + 2], F[FH_IN_BASIS + 3]};
  const uint32_t out_basis[4] = {F[FH_OUT_BASIS], F[FH_OUT_BASIS + 1], F[FH_OUT_BASIS + 2], F[FH_OUT_BASIS + 3]};
  // byte offsets xor-ed into a thread's base offset for the 16 elements it loads (bits of the element number)
  const uint32_t in_step[4] = {direct_in ? in_basis[0] : 16u << ipos[0], direct_in ? in_basis[1] : 16u << ipos[1],
                               direct_in ? in_basis[2] : 16u << ipos[2], direct_in ? in_basis[3] : 16u << ipos[3]};
  // thread parts of the ordinary tile <-> HBM maps, computed once per workgroup and parked in LDS (recomputed per
  // tile they cost ~500 VALU instructions per wave and tile; kept in registers they were spilled)
  tw_slots[t] = xor_map16(t, kt, in_mask) | (xor_map16(t, kt, out_mask) << 16);
  if (!direct_in) tw_in[t] = ((t & ((1u << lo_in) - 1u)) | deposit16(t, lo_in, kt, in_phys)) << 4;
  if (!direct_out) tw_out[t] = xor_cols(t, kt, P + PW_OUT_COL) << out_shift;      // phys-out address: GF(2)-linear (PW_OUT_COL)
  double2* __restrict__ mats_b = tile + lds_mats2_off;   // second matrix buffer: the NEXT tile's matrices land here

  d2_t v[MAX_TILE_ITERS];   // amplitudes of the NEXT tile (in flight during the current tile's stages)
  d2_t mp;                  // its piece of the matrices
#pragma unroll
  for (int i = 0; i < MAX_TILE_ITERS; ++i) v[i] = (d2_t){0.0, 0.0};
  mp = (d2_t){0.0, 0.0};
  uint32_t g_pref = 0xffffffffu;   // tile row whose first-stage offsets are in tw_in (direct_in)
#define BORNVI_PREFETCH(Tn)                                                                          \
  do {                                                                                               \
    const uint32_t gn_ = (uint32_t)((Tn) & ((1ll << gbits) - 1));                                    \
    const long long bn_ = (Tn) >> gbits;                                                             \
    const double* gsrc_ = gates + bn_ * gate_stride;                                                 \
    uint32_t tt_ = t;                                                                                \
    asm volatile("" : "+v"(tt_));   /* (nothing derived from the thread id is hoisted out of the loop) */ \
    if (direct_in && gn_ != g_pref) {   /* rare: compiler-tracked load, waited for inside this branch */ \
      g_pref = gn_;                                                                                  \
      tw_in[tt_] = fast[in_tab + (gn_ << kt) + tt_];                                                 \
    }                                                                                                \
    if (tt_ < npieces && !next_zero) async_load16(mp, tw_mat[tt_], gsrc_);                           \
    if (!init && !(dbg & 4)) {                                                                       \
      /* one per-thread base offset, 16 wave-uniform offsets xor-ed in (both maps are bitwise disjoint or   \
         GF(2)-linear); the empty asm keeps the 16 sums from being hoisted out of the tile loop and spilled */ \
      const uint32_t base_ = tw_in[tt_];                                                             \
      /* (a circuit entering the batch in this pass starts from the base circuit's state: slot 0) */ \
      const double2* src_ = in + (bn_ >= share.fresh_begin ? 0ll : bn_) * state_stride +             \
                            (direct_in ? 0u : deposit16(gn_, 0, gbits, in_gphys));                  \
      _Pragma("unroll") for (int i = 0; i < MAX_TILE_ITERS; ++i)                                     \
        if (!((zslots >> i) & 1u))                                                                   \
          async_load16(v[i], base_ ^ (((i & 1) ? in_step[0] : 0u) ^ ((i & 2) ? in_step[1] : 0u) ^    \
                                      ((i & 4) ? in_step[2] : 0u) ^ ((i & 8) ? in_step[3] : 0u)), src_); \
    }                                                                                                \
  } while (0)

  // ---- the tile loop.  There is exactly ONE expansion of the prefetch in the kernel, inside the loop: trip -1 only
  // starts the pipeline (loads the first tile and its matrices), every later trip consumes the tile in flight, issues
  // the loads of the next one and works on its own.  With a second load site (a prologue) the compiler has to merge
  // two definitions of the in-flight registers at the loop header and may do it with register copies placed right
  // behind a load -- copies of registers whose data has not arrived (tools/check_async_regs.py looks for that).
#if BORNVI_STAMPS
  unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long stamp_last = __builtin_readcyclecounter();
#endif
  uint32_t g_tab = 0xffffffffu;   // tile row whose stage tables are in LDS
  uint32_t parity = 1;            // matrix buffer of the current tile (trip -1 stages tile 0's matrices into buffer 0)
  const uint32_t* __restrict__ FS0 = F + FH_WORDS;
#define BORNVI_RUN_STAGE(S_, IO_)                                                                               \
  do {                                                                                                          \
    const uint32_t* __restrict__ FS_ = FS0 + (S_) * FS_WORDS;                                                   \
    uint32_t G_[10];                                                                                            \
    _Pragma("unroll") for (int i_ = 0; i_ < 10; ++i_) G_[i_] = FS_[i_];                                         \
    const uint32_t kind_ = FS_[FS_KIND];   /* fused gates (register bits 0 .. ng-1) | pre sign << 3 | post sign << 4 */ \
    const uint32_t rw_ = tab_rw[(uint32_t)(S_) * T + t];                                                        \
    const uint32_t sg_ = (kind_ >> 3) ? tab_sg[(uint32_t)__popc(sign_any & ((1u << (S_)) - 1u)) * T + t] : 0u;  \
    stage_dispatch<IO_, DEBUG, SPLIT>(kind_, tile, mats + (S_) * 16, rw_, sg_, G_, dbg, v, (IO_) == 2 ? tw_out[t] : 0u, \
                               out_basis, hbm_base, fin, FS_[FS_CROSS] != 0u);                                  \
  } while (0)
  // (direct_mask bit 2: this launch walks the tiles from the last to the first -- alternate passes in opposite directions
  // start on the states the previous pass wrote last, i.e. on what the memory-side cache still holds)
  const long long walk_flip = total_tiles - 1;
  const bool walk_rev = (direct_mask & 4) != 0;
  const bool zskip = BORNVI_SKIP_ZERO_TILES && !DEBUG && init && gbits > 0 && total_tiles < (1ll << 31);
  // direct_mask bit 3: the launcher vouches that pass 0 and pass 1 of this batch both run with the direct first stage
  // on, so the masks of FH_ZINFO (plan.cpp: support of |0..0>) hold: an INIT pass leaves out the tiles nobody will
  // read (zgmask), the pass behind it does not load the slots known to be zero (zslots; v[] is zero from the start of
  // the kernel and those registers are never loaded, so the first stage finds zeros there)
  const uint32_t zinfo = (BORNVI_SKIP_ZERO_TILES && !DEBUG && (direct_mask & 8)) ? F[FH_ZINFO] : 0u;
  const uint32_t zgmask = zskip ? zinfo : 0u;
  const uint32_t zslots = (!init && direct_in) ? (zinfo & 0xffffu) : 0u;
  const uint32_t zs_nb = (uint32_t)(total_tiles >> gbits), zs_gm1 = (1u << gbits) - 1u;   // circuits; zero tiles per circuit
  for (long long Scur = (long long)blockIdx.x - (long long)gridDim.x;; Scur += gridDim.x, parity ^= 1u) {
    const bool real = Scur >= 0;
    const long long Snext = Scur + gridDim.x;
    const bool has_next = Snext < total_tiles;
    // INIT passes with zero tiles (see zero_tile below): the one non-zero tile of every circuit first, spread over all
    // workgroups (with the plain order workgroup w only ever sees tile w mod 2^gbits: an eighth of the workgroups
    // would do all the work), then the zero tiles
    long long Tcur, Tnext;
    if (zskip) {
      const uint32_t sc = (uint32_t)(real ? Scur : 0), sn = (uint32_t)(has_next ? Snext : 0);
      const uint32_t bc_ = sc < zs_nb ? sc : (sc - zs_nb) / zs_gm1, bn_ = sn < zs_nb ? sn : (sn - zs_nb) / zs_gm1;
      Tcur = sc < zs_nb ? ((long long)sc << gbits) : (((long long)bc_ << gbits) | (1u + (sc - zs_nb - bc_ * zs_gm1)));
      Tnext = sn < zs_nb ? ((long long)sn << gbits) : (((long long)bn_ << gbits) | (1u + (sn - zs_nb - bn_ * zs_gm1)));
    } else {
      Tcur = walk_rev ? walk_flip - Scur : Scur;       // (only used where `real`)
      Tnext = walk_rev ? walk_flip - Snext : Snext;    // (only used where `has_next`)
    }
    if (!real && !has_next) break;
    const uint32_t g = real ? (uint32_t)(Tcur & ((1ll << gbits) - 1)) : 0u;
    const long long b = real ? (Tcur >> gbits) : 0;
    // |0..0> lies in tile 0 of its circuit, and the gates of the first pass act inside a tile (a CNOT controlled by a
    // tile-index wire permutes the tile): every other tile of an INIT pass is zero before and after -- no fill, no
    // stages, 16 stores of zeros (7 of 8 tiles at n = 16, 127 of 128 at n = 20)
    const bool zero_tile = zskip && real && g != 0u;
    const bool noop_tile = zero_tile && (g & zgmask) != 0u;          // nobody reads this tile: not even the zeros are written
    const bool next_zero = zskip && has_next && (Tnext & ((1ll << gbits) - 1)) != 0;   // (needs no matrices)
    double2* __restrict__ mats = parity ? mats_b : tile + ksize;         // this tile's matrices
    double2* __restrict__ mats_next = parity ? tile + ksize : mats_b;
    double2* dst = out + b * state_stride;
    double* pdst = probs + (b << n);
    if (fin && share.row_map) {               // output row of circuit b (rows keep the caller's order; < 0: not wanted)
      const int row = share.row_map[b];
      pdst = row < 0 ? share.trash : probs + ((long long)row << n);
    }
    void* hbm_base = fin ? (void*)pdst : (void*)dst;
    BORNVI_STAMP(real ? 6 : 7);     // loop overhead / the pipeline-start trip
    if (real) {
      // ---- the tile has arrived in registers: all but this wave's 16 tile-out stores are done (after trip -1
      // nothing is outstanding) ----
      if (DEBUG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      BORNVI_STAMP(0);              // waiting for the prefetched tile
#pragma unroll
      for (int i = 0; i < MAX_TILE_ITERS; ++i) asm volatile("" : "+v"(v[i]));
      // ---- registers -> LDS: the tile (head CNOTs of the pass folded into the slot), unless the first stage
      // takes the registers as they are ----
      if (init) {
        if (!zero_tile)
          for (uint32_t u = t; u < ksize; u += T) tile[u] = make_double2((u == 0 && g == 0) ? 1.0 : 0.0, 0.0);
      } else if (!direct_in && !(dbg & 4)) {
        const uint32_t slot_t = (tw_slots[t] & 0xffffu) ^ xor_map16(g, gbits, in_gmask);
#pragma unroll
        for (int i = 0; i < MAX_TILE_ITERS; ++i) {
          const uint32_t its = ((i & 1) ? imask[0] : 0u) ^ ((i & 2) ? imask[1] : 0u) ^ ((i & 4) ? imask[2] : 0u) ^ ((i & 8) ? imask[3] : 0u);
          tile[slot_t ^ its] = make_double2(v[i].x, v[i].y);
        }
      }
      if (direct_in) BORNVI_RUN_STAGE(0, 1);    // (the matrices were staged a trip ago)
      asm volatile("" ::: "memory");
    }
    BORNVI_STAMP(1);                // tile -> LDS (or the direct first stage)
    if (SPLIT) {
    // ---- the registers are free: the next tile starts its trip from HBM.  Large tiles (one workgroup of >= 4 waves per
    // CU, all in step between barriers): the 16 loads are issued in two halves BETWEEN the stages, and the two halves of
    // the workgroup's waves (SIMD partners: wave w and w + nwaves/2 share a SIMD) issue theirs in alternate stages.  A
    // wave-level 1-KiB load takes ~100 cycles of the CU's memory pipe, so 16 of them issued back to back by all waves at
    // once blocked every wave for 11 % of a trip at n = 20 with the VALU idle (in-kernel stamps,
    // tools/probes/stamp_probe.py); now one partner computes while the other issues: -4 % at n = 20, L = 8, rows
    // bit-identical.  2^11 tiles (four independent workgroups per CU already interleave their phases; measured
    // neutral to +2 %) issue all 16 before the first stage as before.  Either way there is ONE load site per in-flight
    // register (inside the loop below, which always runs at least once) and all loads precede the trip's 16 stores:
    // the hand-counted vmcnt waits are unchanged.
    uint32_t gn_h = 0;
    long long bn_h = 0;
    if (has_next) {
      gn_h = (uint32_t)(Tnext & ((1ll << gbits) - 1));
      bn_h = Tnext >> gbits;
      uint32_t tt_ = t;
      asm volatile("" : "+v"(tt_));
      if (direct_in && gn_h != g_pref) {   /* rare: compiler-tracked load, waited for inside this branch */
        g_pref = gn_h;
        tw_in[tt_] = fast[in_tab + (gn_h << kt) + tt_];
      }
      if (tt_ < npieces && !next_zero) async_load16(mp, tw_mat[tt_], gates + bn_h * gate_stride);
    }
    const bool want_v = has_next && !init && !(dbg & 4);
    BORNVI_STAMP(2);                // head of the prefetch (matrix piece)
    {
      const int s0 = (real && direct_in) ? 1 : 0;
      const int ns = (real && !zero_tile) ? nstages : 0;
      const int niter = (ns - s0 > 0) ? ns - s0 : 1;
#ifndef BORNVI_SPLIT_STAGGER
#define BORNVI_SPLIT_STAGGER 1
#endif
      const int wc = (BORNVI_SPLIT_STAGGER && 2u * (t >> 6) >= (T >> 6)) ? 1 : 0;     // second half of the workgroup's waves
      for (int j = 0; j < niter; ++j) {
        const int s = s0 + j;
        const bool last = j == niter - 1;
#define BORNVI_PREFETCH_HALF(C_)                                                                      \
        if (want_v && (j == wc + 2 * (C_) || (last && wc + 2 * (C_) > j))) {                          \
          uint32_t tt_ = t;                                                                          \
          asm volatile("" : "+v"(tt_));                                                              \
          const uint32_t base_ = tw_in[tt_];                                                         \
          const double2* src_ = in + (bn_h >= share.fresh_begin ? 0ll : bn_h) * state_stride +       \
                                (direct_in ? 0u : deposit16(gn_h, 0, gbits, in_gphys));             \
          _Pragma("unroll") for (int i = 8 * (C_); i < 8 * (C_) + 8; ++i)                            \
            if (!((zslots >> i) & 1u))   /* (wave-uniform: a slot known to hold zeros is not loaded) */ \
              async_load16(v[i], base_ ^ (((i & 1) ? in_step[0] : 0u) ^ ((i & 2) ? in_step[1] : 0u) ^  \
                                          ((i & 4) ? in_step[2] : 0u) ^ ((i & 8) ? in_step[3] : 0u)), src_); \
        }
        BORNVI_PREFETCH_HALF(0)
        BORNVI_PREFETCH_HALF(1)
#undef BORNVI_PREFETCH_HALF
        if (j == 0 && real) __syncthreads();    // the tile (or the first stage's result) is in LDS
        if (s < ns) {
          if (s == ns - 1 && direct_out) {
            BORNVI_RUN_STAGE(s, 2);
          } else {
            BORNVI_RUN_STAGE(s, 0);
#if BORNVI_TIMING_NO_STAGE_BARRIER   /* experiment (wrong results): what the workgroup-wide barrier between stages costs */
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
            __syncthreads();
#endif
          }
        }
      }
    }
    } else {
    // ---- the registers are free: the next tile starts its trip from HBM now (the only load site) ----
    if (has_next) BORNVI_PREFETCH(Tnext);
    BORNVI_STAMP(2);                // issuing the prefetch
    if (real) {
      __syncthreads();                          // the tile (or the first stage's result) is in LDS
      for (int s = direct_in ? 1 : 0; s < (zero_tile ? 0 : nstages); ++s) {
        if (s == nstages - 1 && direct_out) {
          BORNVI_RUN_STAGE(s, 2);
        } else {
          BORNVI_RUN_STAGE(s, 0);
          __syncthreads();
        }
      }
    }
    }
    if (real) {
      BORNVI_STAMP(3);              // the stages (LDS round trips, gates, barriers)
      // ---- tile out: LDS -> HBM in the next pass's bit order, or |psi|^2 in canonical order: exactly 16
      // vector-memory stores per wave (the vmcnt waits count them), here or in the last stage ----
      if (noop_tile) {
      } else if (zero_tile) {
        // the same 16 stores per wave as any other tile (the vmcnt waits count them), of zeros, at the tile drain's addresses
        const uint32_t gout0 = xor_cols(g, gbits, P + PW_OUT_GCOL) << out_shift;
        const uint32_t thr0 = xor_cols(t, kt, P + PW_OUT_COL) << out_shift;     // (tw_out may hold a direct stage's row)
#pragma unroll
        for (int i = 0; i < MAX_TILE_ITERS; ++i) {
          const uint32_t off = thr0 ^ gout0 ^ ((i & 1) ? pcol[0] : 0u) ^ ((i & 2) ? pcol[1] : 0u) ^ ((i & 4) ? pcol[2] : 0u) ^ ((i & 8) ? pcol[3] : 0u);
          if (fin) async_store8(off, 0.0, pdst);
          else async_store16(off, (d2_t){0.0, 0.0}, dst);
        }
      } else if (!direct_out) {
        const uint32_t gout = xor_cols(g, gbits, P + PW_OUT_GCOL) << out_shift;
        const uint32_t thr_l = (tw_slots[t] >> 16) ^ xor_map16(g, gbits, out_gmask);   // tail CNOTs folded in
        const uint32_t thr_out = tw_out[t];
#pragma unroll
        for (int i = 0; i < MAX_TILE_ITERS; ++i) {
          const uint32_t it_l = ((i & 1) ? lpos[0] : 0u) ^ ((i & 2) ? lpos[1] : 0u) ^ ((i & 4) ? lpos[2] : 0u) ^ ((i & 8) ? lpos[3] : 0u);
          const uint32_t it_p = gout ^ ((i & 1) ? pcol[0] : 0u) ^ ((i & 2) ? pcol[1] : 0u) ^ ((i & 4) ? pcol[2] : 0u) ^ ((i & 8) ? pcol[3] : 0u);
          const double2 x = tile[thr_l ^ it_l];
          if (DEBUG && (dbg & 8)) continue;
          if (fin) async_store8(thr_out ^ it_p, x.x * x.x + x.y * x.y, pdst);
          else async_store16(thr_out ^ it_p, (d2_t){x.x, x.y}, dst);
        }
      }
    }
    BORNVI_STAMP(4);                // tile out (LDS reads + store issue)
    if (!has_next) break;
    // ---- stage tables of the NEXT tile's row -> LDS.  The launcher makes the grid a multiple of the tiles per
    // state whenever it can, so a workgroup keeps its row and this runs once, in trip -1, beside the first prefetch
    // (these are compiler-tracked loads: they drain vmcnt).  All stages of the current tile are done here. ----
    {
      const uint32_t gnx = (uint32_t)(Tnext & ((1ll << gbits) - 1));
      if (gnx != g_tab && !(zskip && gnx != 0u)) {     // (a zero tile needs no tables)
        g_tab = gnx;
        uint32_t tt = t;
        asm volatile("" : "+v"(tt));
        const uint32_t row = (gnx << kt) + tt;
        for (int s = 0; s < nstages; ++s) {
          tab_rw[(uint32_t)s * T + t] = fast[rw_base + (uint32_t)s * stage_words + row];
          if ((sign_any >> s) & 1u)
            tab_sg[(uint32_t)__popc(sign_any & ((1u << s) - 1u)) * T + t] = fast[sg_base + (uint32_t)s * stage_words + row];
        }
        if (direct_out) tw_out[tt] = fast[out_tab + row];
      }
    }
    // ---- the next tile's matrices (the oldest loads in flight: everything but the 16 amplitude loads and the 16
    // stores behind them is done after vmcnt(32); trip -1 and INIT passes have fewer ops in flight) -> the other buffer ----
    if (DEBUG || !real) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (init || zslots) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // (fewer than 16 loads in flight: all but the stores)
    else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    asm volatile("" : "+v"(mp));
    if (t < npieces) mats_next[t] = make_double2(mp.x, mp.y);
    __syncthreads();   // the tile is overwritten by the next trip; its matrices are in place
    BORNVI_STAMP(5);                // end of trip: wait for the matrices in flight, stage them, barrier
  }
#if BORNVI_STAMPS
  if (threadIdx.x == 0) {
    for (int ph = 0; ph < 8; ++ph) atomicAdd(&g_stamp_totals[ph], stamp_acc[ph]);
    atomicAdd(&g_stamp_totals[8], 1ull);
  }
#endif
#undef BORNVI_RUN_STAGE
#undef BORNVI_PREFETCH
}

// ------------------------------------------------------------------------------------------------
// un-fused kernels: one gate = one HBM round trip (32 * 2^n bytes per state); flat index over
// [batch * 2^n], physical bit of wire w is n-1-w so pairs never straddle two states.
// ------------------------------------------------------------------------------------------------
struct Mat2 { double m[8]; };

__global__ __launch_bounds__(256) void gate1q_kernel(double2* __restrict__ state, long long npairs, int pbit, Mat2 U) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long lowmask = (1ll << pbit) - 1;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < npairs; idx += stride) {
    const long long i0 = ((idx & ~lowmask) << 1) | (idx & lowmask);
    const long long i1 = i0 | (1ll << pbit);
    const double2 x0 = state[i0], x1 = state[i1];
    double2 y0, y1;
    y0.x = U.m[0] * x0.x - U.m[1] * x0.y + U.m[2] * x1.x - U.m[3] * x1.y;
    y0.y = U.m[0] * x0.y + U.m[1] * x0.x + U.m[2] * x1.y + U.m[3] * x1.x;
    y1.x = U.m[4] * x0.x - U.m[5] * x0.y + U.m[6] * x1.x - U.m[7] * x1.y;
    y1.y = U.m[4] * x0.y + U.m[5] * x0.x + U.m[6] * x1.y + U.m[7] * x1.x;
    state[i0] = y0;
    state[i1] = y1;
  }
}

__global__ __launch_bounds__(256) void cnot_kernel(double2* __restrict__ state, long long npairs, int cbit, int tbit) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long lowmask = (1ll << tbit) - 1;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < npairs; idx += stride) {
    const long long i0 = ((idx & ~lowmask) << 1) | (idx & lowmask);
    if (!((i0 >> cbit) & 1)) continue;
    const long long i1 = i0 | (1ll << tbit);
    const double2 x0 = state[i0], x1 = state[i1];
    state[i0] = x1;
    state[i1] = x0;
  }
}

__global__ __launch_bounds__(256) void born_probs_kernel(const double2* __restrict__ state, double* __restrict__ probs, long long total) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const double2 v = state[i];
    probs[i] = v.x * v.x + v.y * v.y;
  }
}

// grad[p] = scale * sum_z w[z] (q+[z] - q-[z]);  rows (+p, -p) of `shifted`; one workgroup per p.
// scale = 1/2 (ksd2 == nullptr) or 1/2 / sqrt(max(ksd2, 1e-12)) with the clamp's zero gradient
// below 1e-12 (ksd_vi_quantum.py:145).  Fixed-order tree reduction: deterministic.
constexpr int SHIFT_DOT_THREADS = 1024;
__global__ __launch_bounds__(SHIFT_DOT_THREADS) void shift_dot_kernel(const double* __restrict__ shifted, const double* __restrict__ w,
                                                                     const double* __restrict__ ksd2, long long N,
                                                                     double* __restrict__ grad, double* __restrict__ loss_out) {
  __shared__ double red[SHIFT_DOT_THREADS / 64];
  const long long p = blockIdx.x;
  const double* qp = shifted + (2 * p) * N;
  const double* qm = qp + N;
  double acc = 0.0;
  if ((N & 1) == 0) {
    // 16-byte loads, four pairs in flight per thread: one workgroup (16 waves) per parameter streams its two
    // probability vectors at the CU's full rate
    const long long N2 = N >> 1;
    // (the two rows are read once: non-temporal loads; w is shared by every workgroup and stays cached)
    const d2_t* __restrict__ qp2 = reinterpret_cast<const d2_t*>(qp);
    const d2_t* __restrict__ qm2 = reinterpret_cast<const d2_t*>(qm);
    const double2* __restrict__ w2 = reinterpret_cast<const double2*>(w);
    long long z = threadIdx.x;
    for (; z + 3 * SHIFT_DOT_THREADS < N2; z += 4 * SHIFT_DOT_THREADS) {
      d2_t a[4], b[4];
      double2 c[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = __builtin_nontemporal_load(qp2 + z + u * SHIFT_DOT_THREADS);
        b[u] = __builtin_nontemporal_load(qm2 + z + u * SHIFT_DOT_THREADS);
        c[u] = w2[z + u * SHIFT_DOT_THREADS];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc += c[u].x * (a[u].x - b[u].x) + c[u].y * (a[u].y - b[u].y);
    }
    for (; z < N2; z += SHIFT_DOT_THREADS) {
      const d2_t a = __builtin_nontemporal_load(qp2 + z), b = __builtin_nontemporal_load(qm2 + z);
      const double2 c = w2[z];
      acc += c.x * (a.x - b.x) + c.y * (a.y - b.y);
    }
  } else {
    for (long long z = threadIdx.x; z < N; z += SHIFT_DOT_THREADS) acc += w[z] * (qp[z] - qm[z]);
  }
  // wave shuffles, then the 16 wave sums in fixed order: deterministic
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < SHIFT_DOT_THREADS / 64; ++w) tot += red[w];
    red[0] = tot;
  }
  if (threadIdx.x == 0) {
    double scale = 0.5;
    if (ksd2) {
      const double k2 = *ksd2;
      const double loss = sqrt(k2 < 1e-12 ? 1e-12 : k2);
      scale = (k2 < 1e-12) ? 0.0 : 0.5 / loss;
      if (p == 0 && loss_out) *loss_out = loss;
    }
    grad[p] = scale * red[0];
  }
}

__global__ __launch_bounds__(256) void dldq_kernel(const double* __restrict__ y, const double* __restrict__ ksd2,
                                                   long long N, double* __restrict__ dLdq, double* __restrict__ loss_out) {
  const double k2 = *ksd2;
  const double loss = sqrt(k2 < 1e-12 ? 1e-12 : k2);
  const double inv = (k2 < 1e-12) ? 0.0 : 1.0 / loss;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride)
    if (dLdq) dLdq[i] = y[i] * inv;
  if (blockIdx.x == 0 && threadIdx.x == 0 && loss_out) *loss_out = loss;
}

// Gradient hand-off to the optimiser in one launch: float32 cast of the float64 gradient, its 2-norm, and the
// clip of torch.nn.utils.clip_grad_norm_ (coef = min(1, max_norm / (norm + 1e-6)); ksd_vi_quantum.py:153).
// loss / found_inf (both or neither): found_inf = 1.0f when the loss is NaN or +-Inf (the reference skips the update
// then, ksd_vi_quantum.py:147-148; torch's fused optimisers take this flag on the device), else 0.0f.
__global__ __launch_bounds__(256) void clip_cast_kernel(const double* __restrict__ g64, int P, double max_norm,
                                                        float* __restrict__ g32, float* __restrict__ norm_out,
                                                        const double* __restrict__ loss, float* __restrict__ found_inf) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < P; i += 256) { const float f = (float)g64[i]; acc += (double)f * (double)f; }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  const float total = (float)sqrt(red[0]);
  float coef = (float)max_norm / (total + 1e-6f);
  if (coef > 1.0f) coef = 1.0f;
  for (int i = threadIdx.x; i < P; i += 256) g32[i] = (float)g64[i] * coef;
  if (threadIdx.x == 0) {
    *norm_out = total;
    if (found_inf) { const double l = *loss; *found_inf = (l - l == 0.0) ? 0.0f : 1.0f; }   // l - l is NaN for NaN and +-Inf
  }
}

hipError_t launch_clip_cast(const double* g64, int P, double max_norm, float* g32, float* norm_out, const double* loss,
                            float* found_inf, hipStream_t st) {
  clip_cast_kernel<<<1, 256, 0, st>>>(g64, P, max_norm, g32, norm_out, loss, found_inf);
  return hipGetLastError();
}

// The whole optimiser hand-off of one epoch in ONE launch: clip_cast_kernel's norm / clip / NaN-Inf guard followed by the
// Adam update of torch.optim.Adam (ksd_vi_quantum.py:92-99: no weight decay, no amsgrad) on float32 theta, in the
// arithmetic of torch's single-kernel implementation (moments and parameter in float32, the products with the double
// hyper-parameters in double).  For the latency-bound sizes: a replayed step spent 9 of its 15 graph nodes in torch's
// optimiser, schedule and cast kernels.
//   counters[0] = good steps so far (the bias corrections' step count; a skipped epoch does not advance it),
//   counters[1] = epochs so far (indexes lr_table, clamped to its last entry; advances on every call: the reference's
//                 scheduler steps on a skipped epoch as well, ksd_vi_quantum.py:158-160).
// loss_hist / norm_hist (or null): the same slot receives the epoch's loss and gradient norm -- the history the
// reference appends per epoch (:163-166), without a copy launch per epoch.
// theta64: the float64 copy of the updated theta the next epoch's circuits read (saves the cast launch).
__global__ __launch_bounds__(256) void clip_adam_kernel(const double* __restrict__ g64, int P, double max_norm,
                                                        const double* __restrict__ loss, float* __restrict__ theta,
                                                        float* __restrict__ g32, double* __restrict__ theta64,
                                                        float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq,
                                                        int* __restrict__ counters, const double* __restrict__ lr_table,
                                                        int n_lr, double beta1, double beta2, double eps,
                                                        float* __restrict__ norm_out, double* __restrict__ loss_hist,
                                                        float* __restrict__ norm_hist) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < P; i += 256) { const float f = (float)g64[i]; acc += (double)f * (double)f; }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  const float total = (float)sqrt(red[0]);
  float coef = (float)max_norm / (total + 1e-6f);
  if (coef > 1.0f) coef = 1.0f;
  const double l = *loss;
  const bool good = (l - l == 0.0);
  const int step = counters[0] + (good ? 1 : 0), epoch = counters[1];
  const int slot = epoch < n_lr ? epoch : n_lr - 1;
  const double lr = lr_table[slot];
  const float bc1 = (float)(1.0 - pow(beta1, (double)step));
  const float bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
  const float step_size = (float)(lr / (double)bc1);
  __syncthreads();                       // every thread has read the counters
  for (int i = threadIdx.x; i < P; i += 256) {
    const float g = (float)g64[i] * coef;
    g32[i] = g;
    float p = theta[i];
    if (good) {
      const float m = (float)(beta1 * (double)exp_avg[i] + (1.0 - beta1) * (double)g);
      const float v = (float)(beta2 * (double)exp_avg_sq[i] + (1.0 - beta2) * (double)g * (double)g);
      const float denom = (float)((double)(sqrtf(v) / bc2_sqrt) + eps);
      p -= step_size * m / denom;
      exp_avg[i] = m;
      exp_avg_sq[i] = v;
      theta[i] = p;
    }
    theta64[i] = (double)p;
  }
  if (threadIdx.x == 0) {
    *norm_out = total;
    if (loss_hist) loss_hist[slot] = l;          // the epoch's history entries (no copy launches per epoch)
    if (norm_hist) norm_hist[slot] = total;
    counters[0] = step;
    counters[1] = epoch + 1;
  }
}

hipError_t launch_clip_adam(const double* g64, int P, double max_norm, const double* loss, float* theta, float* g32,
                            double* theta64, float* exp_avg, float* exp_avg_sq, int* counters, const double* lr_table,
                            int n_lr, double beta1, double beta2, double eps, float* norm_out, double* loss_hist,
                            float* norm_hist, hipStream_t st) {
  clip_adam_kernel<<<1, 256, 0, st>>>(g64, P, max_norm, loss, theta, g32, theta64, exp_avg, exp_avg_sq, counters, lr_table,
                                      n_lr, beta1, beta2, eps, norm_out, loss_hist, norm_hist);
  return hipGetLastError();
}

// ---- launchers (called from api.hip) --------------------------------------------------------------------
hipError_t launch_build_gates(const uint32_t* plan, int nfused, const double* thetas, long long theta_stride,
                              int shift_mode, int p_begin, int p_stride, int include_base, long long b_offset, int batch,
                              double* gates, const int* shift_tab, int slots, int normalise, hipStream_t st) {
  const long long total = (long long)batch * nfused;
  if (total == 0) return hipSuccess;
  const int bs = 128;
  build_gates_kernel<<<dim3((unsigned)((total + bs - 1) / bs)), dim3(bs), 0, st>>>(
      plan, thetas, theta_stride, shift_mode, p_begin, p_stride, include_base, b_offset, batch, gates, shift_tab, slots, normalise);
  // (normalise == 2: a single-pass plan -- the pass kernel multiplies the pivots itself, one launch less)
  if (normalise == 1) gate_scale_kernel<<<dim3((unsigned)batch), dim3(64), 0, st>>>(gates, nfused, slots, batch);
  return hipGetLastError();
}

hipError_t launch_normalise_gates(double* gates, int count, hipStream_t st) {
  if (count <= 0) return hipSuccess;
  normalise_gates_kernel<<<dim3((unsigned)((count + 63) / 64)), dim3(64), 0, st>>>(gates, count);
  return hipGetLastError();
}

hipError_t prepare_circuit_kernel(size_t lds_bytes) {
  const void* fns[6] = {reinterpret_cast<const void*>(circuit_pass_kernel<true, false>),
                        reinterpret_cast<const void*>(circuit_pass_kernel<false, false>),
                        reinterpret_cast<const void*>(circuit_pass_kernel<true, true>),
                        reinterpret_cast<const void*>(circuit_pass_kernel<false, true>),
                        reinterpret_cast<const void*>(circuit_pass_fast_kernel<false, false>),
                        reinterpret_cast<const void*>(circuit_pass_fast_kernel<true, false>)};
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(circuit_pass_fast_kernel<false, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
  }
  for (const void* f : fns) {
    hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

int circuit_fast_workgroups_per_cu(int threads, size_t lds) {
  int nb = 0;
  const hipError_t e = (BORNVI_SPLIT_PREFETCH && threads >= 256)
                           ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, circuit_pass_fast_kernel<false, true>, threads, lds)
                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, circuit_pass_fast_kernel<false, false>, threads, lds);
  if (e != hipSuccess) return 0;
  return nb;
}

hipError_t launch_circuit_pass_fast(const uint32_t* plan, uint32_t pass_off, const uint32_t* fast, uint32_t fast_off,
                                    int n, int k, size_t lds, int batch, const void* in, void* out, double* probs,
                                    const double* gates, long long gate_stride, int max_workgroups, size_t lds_tab_off,
                                    size_t lds_mats2_off, int direct_mask, int dbg, const PrefixShare& share, hipStream_t st) {
  const long long total_tiles = (long long)batch << (n - k);
  if (total_tiles == 0) return hipSuccess;
  long long wgs = (max_workgroups > 0 && total_tiles > max_workgroups) ? max_workgroups : total_tiles;
  // a grid that is a multiple of the tiles per state keeps every workgroup on one tile row (its stage tables
  // stay in LDS for the whole launch)
  const long long per_state = 1ll << (n - k);
  if (wgs > per_state) wgs -= wgs % per_state;
  dim3 grid((unsigned)wgs);
  const dim3 block(1u << (k - 4));
#define BORNVI_LAUNCH_FAST(D_, S_, DBG_)                                                                                 \
  circuit_pass_fast_kernel<D_, S_><<<grid, block, lds, st>>>(plan, pass_off, fast, fast_off, (const double2*)in, (double2*)out, \
                                                             probs, gates, gate_stride, 1ll << n, total_tiles,            \
                                                             (uint32_t)(lds_tab_off / 16), (uint32_t)(lds_mats2_off / 16), direct_mask, DBG_, share)
  // large tiles (one workgroup of >= 4 waves per CU): the instantiation that spreads the prefetch over the stages
  if (dbg) BORNVI_LAUNCH_FAST(true, false, dbg);
  else if (BORNVI_SPLIT_PREFETCH && block.x >= 256) BORNVI_LAUNCH_FAST(false, true, 0);
  else BORNVI_LAUNCH_FAST(false, false, 0);
#undef BORNVI_LAUNCH_FAST
  return hipGetLastError();
}

// diagnostic builds: read (and clear) the phase-stamp totals; zeros in the production build
hipError_t read_circuit_stamps(unsigned long long* out16) {
  for (int i = 0; i < 16; ++i) out16[i] = 0;
#if BORNVI_STAMPS
  hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stamp_totals), 16 * sizeof(unsigned long long));
  if (e != hipSuccess) return e;
  unsigned long long zero[16] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_totals), zero, sizeof(zero));
#else
  return hipSuccess;
#endif
}

hipError_t launch_circuit_pass(const uint32_t* plan, uint32_t pass_off, int n, int k, int threads, size_t lds, int batch,
                               const void* in, void* out, double* probs, const double* gates,
                               long long gate_stride, int max_workgroups, int dbg, hipStream_t st) {
  const long long total_tiles = (long long)batch << (n - k);
  dim3 grid((unsigned)((max_workgroups > 0 && total_tiles > max_workgroups) ? max_workgroups : total_tiles));
  int tau = 0;
  while ((1 << tau) < threads) ++tau;
  const bool full = (k - tau) == 4;   // 16 tile elements per thread
#define BORNVI_LAUNCH_PASS(F, D)                                                     \
  circuit_pass_kernel<F, D><<<grid, dim3(threads), lds, st>>>(                       \
      plan, pass_off, (const double2*)in, (double2*)out, probs, gates, gate_stride, 1ll << n, total_tiles, dbg)
  if (full && !dbg) BORNVI_LAUNCH_PASS(true, false);
  else if (full) BORNVI_LAUNCH_PASS(true, true);
  else if (!dbg) BORNVI_LAUNCH_PASS(false, false);
  else BORNVI_LAUNCH_PASS(false, true);
#undef BORNVI_LAUNCH_PASS
  return hipGetLastError();
}

static inline unsigned grid_for(long long work, int bs) {
  long long g = (work + bs - 1) / bs;
  if (g > 256 * 8 * 4) g = 256 * 8 * 4;
  if (g < 1) g = 1;
  return (unsigned)g;
}

hipError_t launch_gate1q(double* state, int n, long long batch, int wire, const double* U, hipStream_t st) {
  Mat2 m;
  for (int i = 0; i < 8; ++i) m.m[i] = U[i];
  const long long npairs = batch << (n - 1);
  gate1q_kernel<<<grid_for(npairs, 256), 256, 0, st>>>((double2*)state, npairs, n - 1 - wire, m);
  return hipGetLastError();
}

hipError_t launch_cnot(double* state, int n, long long batch, int control, int target, hipStream_t st) {
  const long long npairs = batch << (n - 1);
  cnot_kernel<<<grid_for(npairs, 256), 256, 0, st>>>((double2*)state, npairs, n - 1 - control, n - 1 - target);
  return hipGetLastError();
}

hipError_t launch_born_probs(const double* state, double* probs, int n, long long batch, hipStream_t st) {
  const long long total = batch << n;
  born_probs_kernel<<<grid_for(total, 256), 256, 0, st>>>((const double2*)state, probs, total);
  return hipGetLastError();
}

hipError_t launch_shift_dot(const double* shifted, int n_shift, const double* w, const double* ksd2, int n,
                            double* grad, double* loss_out, hipStream_t st) {
  if (n_shift <= 0) return hipSuccess;
  shift_dot_kernel<<<n_shift, SHIFT_DOT_THREADS, 0, st>>>(shifted, w, ksd2, 1ll << n, grad, loss_out);
  return hipGetLastError();
}

hipError_t launch_dldq(const double* y, const double* ksd2, int n, double* dLdq, double* loss_out, hipStream_t st) {
  const long long N = 1ll << n;
  dldq_kernel<<<grid_for(N, 256), 256, 0, st>>>(y, ksd2, N, dLdq, loss_out);
  return hipGetLastError();
}

}  // namespace bornvi
