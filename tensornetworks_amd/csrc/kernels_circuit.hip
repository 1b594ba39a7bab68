// Statevector kernels for gfx950 (MI355X): fused-gate matrices, the LDS-tiled circuit pass,
// the un-fused one-gate kernels (gate-apply micro-benchmark) and Born probabilities.
//
// Data layout: a state is 2^n complex128 (re, im interleaved = double2), outcome index i with
// wire 0 the MOST significant bit (utils.py:77-91 / qml.probs order).  Between passes the
// amplitudes live in HBM in a pass-specific bit permutation chosen by the planner (plan.hpp);
// only the last pass writes the canonical order (as probabilities).
#include <hip/hip_runtime.h>

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
                                   long long theta_stride, int shift_mode, int p_begin, int include_base,
                                   long long b_offset, int batch, double* __restrict__ gates) {
  const uint32_t nf = plan[PH_NFUSED];
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)batch * nf) return;
  const long long b = idx / nf;
  const uint32_t f = (uint32_t)(idx % nf);
  const uint32_t* fw = plan + plan[PH_OFF_FUSED] + f * FUSED_WORDS;
  const long long bg = b + b_offset;
  uint32_t pshift = 0xfffffffeu;
  double shift = 0.0;
  if (shift_mode) {
    const long long bb = bg - include_base;
    if (bb >= 0) { pshift = (uint32_t)(p_begin + (bb >> 1)); shift = (bb & 1) ? -M_PI_2 : M_PI_2; }
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
  double* dst = gates + (b * nf + f) * 8;
#pragma unroll
  for (int i = 0; i < 8; ++i) dst[i] = U[i];
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
    long long gate_stride, long long state_stride, int dbg_arg) {
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
  const uint32_t g = blockIdx.x;
  const long long b = blockIdx.y;
  const uint32_t ksize = 1u << k;
  const int niter = 1 << (k - kt);    // <= MAX_TILE_ITERS (checked by the planner: threads >= 2^(k-4))
  double2* __restrict__ mats = tile + ksize;   // [nstages][4 register bits][4 double2] fused matrices of circuit b

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
    if (t < (1u << (k - r))) {
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
      const int nreg = 1 << r;
      const uint32_t rbase = pb ^ lflip, wbase = pb ^ sflip;
      double ar[16], ai[16];
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
    const uint32_t gout = deposit16(g, 0, n - k, out_gphys);
    const uint32_t thr_l = xor_map16(t, kt, out_mask) ^ xor_map16(g, n - k, out_gmask);   // tail CNOTs folded in
    const uint32_t thr_p = (t & ((1u << lo_out) - 1u)) | deposit16(t, lo_out, kt, out_phys);
    uint32_t lpos[4], ppos[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      lpos[m] = (kt + m < k) ? half16_dyn(out_mask, kt + m) : 0u;
      ppos[m] = (kt + m < k) ? byte16(out_phys, (uint32_t)(kt + m)) : 0u;
    }
    const bool fin = flags & PASS_FINAL;
    double2* __restrict__ dst = out + b * state_stride + gout;
    double* __restrict__ pdst = probs + (b << n) + gout;
    auto move_out = [&](int i) {
      const uint32_t it_l = ((i & 1) ? lpos[0] : 0u) ^ ((i & 2) ? lpos[1] : 0u) ^ ((i & 4) ? lpos[2] : 0u) ^ ((i & 8) ? lpos[3] : 0u);
      const uint32_t it_p = ((i & 1) ? 1u << ppos[0] : 0u) | ((i & 2) ? 1u << ppos[1] : 0u) |
                            ((i & 4) ? 1u << ppos[2] : 0u) | ((i & 8) ? 1u << ppos[3] : 0u);
      const double2 v = tile[thr_l ^ it_l];
      if (fin) pdst[thr_p | it_p] = v.x * v.x + v.y * v.y;
      else dst[thr_p | it_p] = v;
    };
    if (FULL) {
#pragma unroll
      for (int i = 0; i < MAX_TILE_ITERS; ++i) move_out(i);
    } else {
      for (int i = 0; i < niter; ++i) move_out(i);
    }
  }
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
__global__ __launch_bounds__(256) void shift_dot_kernel(const double* __restrict__ shifted, const double* __restrict__ w,
                                                        const double* __restrict__ ksd2, long long N,
                                                        double* __restrict__ grad, double* __restrict__ loss_out) {
  __shared__ double red[256];
  const long long p = blockIdx.x;
  const double* qp = shifted + (2 * p) * N;
  const double* qm = qp + N;
  double acc = 0.0;
  for (long long z = threadIdx.x; z < N; z += blockDim.x) acc += w[z] * (qp[z] - qm[z]);
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
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
__global__ __launch_bounds__(256) void clip_cast_kernel(const double* __restrict__ g64, int P, double max_norm,
                                                        float* __restrict__ g32, float* __restrict__ norm_out) {
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
  if (threadIdx.x == 0) *norm_out = total;
}

hipError_t launch_clip_cast(const double* g64, int P, double max_norm, float* g32, float* norm_out, hipStream_t st) {
  clip_cast_kernel<<<1, 256, 0, st>>>(g64, P, max_norm, g32, norm_out);
  return hipGetLastError();
}

// ---- launchers (called from api.hip) --------------------------------------------------------------------
hipError_t launch_build_gates(const uint32_t* plan, int nfused, const double* thetas, long long theta_stride,
                              int shift_mode, int p_begin, int include_base, long long b_offset, int batch,
                              double* gates, hipStream_t st) {
  const long long total = (long long)batch * nfused;
  if (total == 0) return hipSuccess;
  const int bs = 128;
  build_gates_kernel<<<dim3((unsigned)((total + bs - 1) / bs)), dim3(bs), 0, st>>>(
      plan, thetas, theta_stride, shift_mode, p_begin, include_base, b_offset, batch, gates);
  return hipGetLastError();
}

hipError_t prepare_circuit_kernel(size_t lds_bytes) {
  const void* fns[4] = {reinterpret_cast<const void*>(circuit_pass_kernel<true, false>),
                        reinterpret_cast<const void*>(circuit_pass_kernel<false, false>),
                        reinterpret_cast<const void*>(circuit_pass_kernel<true, true>),
                        reinterpret_cast<const void*>(circuit_pass_kernel<false, true>)};
  for (const void* f : fns) {
    hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

hipError_t launch_circuit_pass(const uint32_t* plan, uint32_t pass_off, int n, int k, int threads, size_t lds, int batch,
                               const void* in, void* out, double* probs, const double* gates,
                               long long gate_stride, int dbg, hipStream_t st) {
  dim3 grid(1u << (n - k), (unsigned)batch);
  int tau = 0;
  while ((1 << tau) < threads) ++tau;
  const bool full = (k - tau) == 4;   // 16 tile elements per thread
#define BORNVI_LAUNCH_PASS(F, D)                                                     \
  circuit_pass_kernel<F, D><<<grid, dim3(threads), lds, st>>>(                       \
      plan, pass_off, (const double2*)in, (double2*)out, probs, gates, gate_stride, 1ll << n, dbg)
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
  shift_dot_kernel<<<n_shift, 256, 0, st>>>(shifted, w, ksd2, 1ll << n, grad, loss_out);
  return hipGetLastError();
}

hipError_t launch_dldq(const double* y, const double* ksd2, int n, double* dLdq, double* loss_out, hipStream_t st) {
  const long long N = 1ll << n;
  dldq_kernel<<<grid_for(N, 256), 256, 0, st>>>(y, ksd2, N, dLdq, loss_out);
  return hipGetLastError();
}

}  // namespace bornvi
