// Persistent circuit pass kernel with 8 amplitudes per thread (3 register wires per stage) for gfx950.
//
// Same plan model, same tile <-> HBM maps and the same hand-counted vector-memory pipeline as circuit_pass_fast_kernel
// (kernels_circuit.hip), re-cut for occupancy: a thread owns 2^3 amplitudes of the tile instead of 2^4,
// so the stage body, the in-flight next tile and a matrix fit 128 VGPRs and FOUR waves per SIMD are resident (a 2^13
// tile is one 1024-thread workgroup per CU, 2^11 tiles four 256-thread workgroups) instead of two.  With twice the threads per tile a
// per-thread table row would be 4 KiB per stage and no longer fit beside the tile, so the tables come in their
// COMPACT form (plan.hpp: CompactTables): every per-(tile row, thread) word is GF(2)-affine in (tile row, thread), i.e.
// word = LANE[row][lane] ^ UNI[row][wave], 64 + 16 words per row in LDS.
// The fused 2x2 matrices are wave-uniform: they come by SCALAR loads straight from the gate array into SGPRs (every
// instruction of the gate reads one matrix element = one SGPR pair, the constant-bus limit of VOP3) -- no LDS staging, no
// broadcast ds_read (which was a third of the LDS traffic of a stage), 16 VGPRs less.
#include <hip/hip_runtime.h>

#include "circuit_dev.hpp"
#include "kernels.hpp"
#include "plan.hpp"

namespace bornvi {

// timing-only ablations (wrong results by construction; tools/probes/build_r3_variants.sh): what the parts of a stage cost
#ifndef BORNVI_R3_NO_MATLOAD
#define BORNVI_R3_NO_MATLOAD 0     // 1: the gates' matrices are not read from LDS
#endif
#ifndef BORNVI_R3_NO_STAGES
#define BORNVI_R3_NO_STAGES 0      // 1: tile in, tile out, nothing in between (the HBM side of a pass alone)
#endif
#ifndef BORNVI_R3_STAGGER
#define BORNVI_R3_STAGGER 0        // experiment: workgroups start 0 .. 3 quarters of this many 64-cycle ticks late (phase spread over the CUs)
#endif
#ifndef BORNVI_R3_MAT_UPFRONT
#define BORNVI_R3_MAT_UPFRONT 1    // 1: the stage's matrices are requested together, ahead of the amplitude reads; 0: each one in front of its gate (A/B)
#endif
#ifndef BORNVI_R3_NO_LDS_READ
#define BORNVI_R3_NO_LDS_READ 0    // 1: a stage does not read its amplitudes from LDS (what the read half of the round trip costs)
#endif
#ifndef BORNVI_R3_NO_LDS_WRITE
#define BORNVI_R3_NO_LDS_WRITE 0   // 1: a stage does not write its results back to LDS
#endif
#ifndef BORNVI_R3_NO_GATES
#define BORNVI_R3_NO_GATES 0       // 1: the stages' LDS round trips and signs without the gate arithmetic
#endif

namespace {

// One gate on an amplitude pair, matrix as the pivot-normalised RECORD build_gates_kernel writes (kernels_circuit.hip):
// R = (B, C, D, |p|^2, exchange flag):  z0 = x0 + B x1,  z1 = C x0 + D x1  -- 12 fp64 instructions instead of the 16 of a
// general complex 2x2 (the pivot's modulus is carried as a scale of the circuit's probabilities, its phase is global).
// In place; every instruction reads one record element = one SGPR pair (the constant-bus limit of VOP3).
__device__ __forceinline__ void gate_pair8(double& x0r, double& x0i, double& x1r, double& x1i, const double (&R)[8]) {
  double t2, t3;
  asm("v_mul_f64 %4, %9, %1\n\t"          // t2 = Ci x0i
      "v_mul_f64 %5, %9, %0\n\t"          // t3 = Ci x0r
      "v_fma_f64 %4, %8, %0, -%4\n\t"     // t2 = Cr x0r - Ci x0i
      "v_fma_f64 %5, %8, %1, %5\n\t"      // t3 = Cr x0i + Ci x0r
      "v_fma_f64 %0, %6, %2, %0\n\t"      // x0r += Br x1r
      "v_fma_f64 %1, %6, %3, %1\n\t"      // x0i += Br x1i
      "v_fma_f64 %0, -%7, %3, %0\n\t"     // x0r -= Bi x1i
      "v_fma_f64 %1, %7, %2, %1\n\t"      // x0i += Bi x1r
      "v_fma_f64 %4, -%11, %3, %4\n\t"    // t2 -= Di x1i
      "v_fma_f64 %5, %11, %2, %5\n\t"     // t3 += Di x1r
      "v_fma_f64 %2, %10, %2, %4\n\t"     // x1r = Dr x1r + t2
      "v_fma_f64 %3, %10, %3, %5"          // x1i = Dr x1i + t3
      : "+v"(x0r), "+v"(x0i), "+v"(x1r), "+v"(x1i), "=&v"(t2), "=&v"(t3)
      : "s"(R[0]), "s"(R[1]), "s"(R[2]), "s"(R[3]), "s"(R[4]), "s"(R[5]));
}
// the record's exchange flag (1.0 / 0.0): the two results of every pair of this gate change places
__device__ __forceinline__ bool rec_swap(const double (&R)[8]) { return __double2hiint(R[7]) != 0; }

// the 2x2 complex matrix at a wave-uniform address: one 64-byte scalar load
__device__ __forceinline__ void load_u8(const double* __restrict__ Um, double (&U)[8]) {
#if BORNVI_R3_NO_MATLOAD
  (void)Um;
#pragma unroll
  for (int e = 0; e < 8; ++e) asm volatile("" : "=s"(U[e]));
#else
#pragma unroll
  for (int e = 0; e < 8; ++e) U[e] = Um[e];
#endif
}

__device__ __forceinline__ uint32_t comb3(int j, const uint32_t (&B)[3]) {
  return ((j & 1) ? B[0] : 0u) ^ ((j & 2) ? B[1] : 0u) ^ ((j & 4) ? B[2] : 0u);
}
__device__ __forceinline__ uint32_t comb3_rt(uint32_t j, const uint32_t (&B)[3]) {     // (wave-uniform j: scalar selects)
  return ((j & 1u) ? B[0] : 0u) ^ ((j & 2u) ? B[1] : 0u) ^ ((j & 4u) ? B[2] : 0u);
}

// xor over bits j in [0, nbits) of v of cols[j] (GF(2)-linear phys-out address, plan.hpp: PW_OUT_COL)
__device__ __forceinline__ uint32_t xor_cols16(uint32_t v, int nbits, const uint32_t* __restrict__ cols) {
  uint32_t o = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j)
    if (j < nbits) o ^= (0u - ((v >> j) & 1u)) & cols[j];
  return o;
}

template <int I>
__device__ __forceinline__ void gate8(double (&ar)[8], double (&ai)[8], const double (&U)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (j & (1 << I)) continue;
    gate_pair8(ar[j], ai[j], ar[j | (1 << I)], ai[j | (1 << I)], U);
  }
}

__device__ __forceinline__ double flip_sign(double x, uint32_t bits, int j) {
  const int sw = (int)((bits << (31 - j)) & 0x80000000u);
  return __hiloint2double(__double2hiint(x) ^ sw, __double2loint(x));
}

__device__ __forceinline__ void sign8(uint32_t m, double (&ar)[8], double (&ai)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) { ar[j] = flip_sign(ar[j], m, j); ai[j] = flip_sign(ai[j], m, j); }
}

// the results of slot jj leave the thread: to LDS (IO 0 / 1), or straight to HBM (IO 2: 16 bytes, or |amp|^2 as 8 bytes)
template <int IO, bool FIN>
__device__ __forceinline__ void put_slot(double xr, double xi, int jj, char* __restrict__ lds, uint32_t wa0, const uint32_t (&WB)[3],
                                         const uint32_t (&hb)[3], void* hbm_base, double scale) {
  if (IO == 2) {
    const uint32_t ha = wa0 ^ comb3(jj, hb);
    if (FIN) async_store8(ha, (xr * xr + xi * xi) * scale, hbm_base);
    else async_store16(ha, (d2_t){xr, xi}, hbm_base);
  } else {
#if BORNVI_R3_NO_LDS_WRITE
    asm volatile("" : : "v"(xr), "v"(xi), "v"(wa0 ^ comb3(jj, WB)));
#else
    *reinterpret_cast<double2*>(lds + (wa0 ^ comb3(jj, WB))) = make_double2(xr, xi);
#endif
  }
}

// last gate of a stage with the write-back folded in: a pair's two results leave right behind its gate (spreads the
// LDS writes / HBM stores over the gate and ends the amplitudes' live ranges early)
template <int I, bool POST, int IO, bool FIN>
__device__ __forceinline__ void gate8_last(double (&ar)[8], double (&ai)[8], const double (&U)[8], uint32_t post_bits,
                                           char* __restrict__ lds, uint32_t wa0, const uint32_t (&WB)[3], const uint32_t (&hb)[3],
                                           void* hbm_base, double scale) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (j & (1 << I)) continue;
    const int j1 = j | (1 << I);
    gate_pair8(ar[j], ai[j], ar[j1], ai[j1], U);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int jj = q ? j1 : j;
      double xr = ar[jj], xi = ai[jj];
      if (POST) { xr = flip_sign(xr, post_bits, jj); xi = flip_sign(xi, post_bits, jj); }
      put_slot<IO, FIN>(xr, xi, jj, lds, wa0, WB, hb, hbm_base, scale);
    }
  }
}

// One stage on the 8 amplitudes a thread owns; NG fused gates on register bits 0 .. NG-1, optional CZ sign products.
// IO: 0 = LDS -> LDS; 1 = the amplitudes are the prefetched registers `v` (first stage of a pass, results to LDS);
// 2 = LDS -> HBM (last stage of a pass).
// DOT (last stage of the final pass, IO == 2 and FIN): no probabilities are stored; acc += sum_j |z_j|^2 * W[slot j's outcome]
// instead -- the parameter-shift dot product with dL/dq fused into the pass (W: the thread's 8 weights, loaded once per
// tile row; a gate's exchange flag permutes which weight a register slot meets: one of eight straight-line variants).
template <int NG, bool PRE, bool POST, int IO, bool FIN, bool DOT = false>
__device__ __forceinline__ void stage8(char* __restrict__ lds, const char* __restrict__ gb, const uint32_t (&MAT)[3], uint32_t my_rw,
                                       uint32_t my_sg, const uint32_t (&RB)[3], const uint32_t (&WB)[3], d2_t (&v)[8], uint32_t hbm_off,
                                       const uint32_t (&hb)[3], void* hbm_base, bool cross, double scale, const double (&W)[8],
                                       double& acc) {
  double ar[8], ai[8];
  // all matrices of the stage are requested before the amplitudes (scalar loads return out of order with LDS reads: one
  // lgkmcnt(0) in front of the first gate covers both)
  // (two register sets: the third matrix is requested into the first set behind gate 0 and arrives under gate 1's FMAs)
  double Ua[8], Ub[8];
#if BORNVI_R3_MAT_UPFRONT
  if (NG > 0) load_u8(reinterpret_cast<const double*>(gb + MAT[0]), Ua);
  if (NG > 1) load_u8(reinterpret_cast<const double*>(gb + MAT[1]), Ub);
#endif
  if (IO == 1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { ar[j] = v[j].x; ai[j] = v[j].y; }
  } else {
    const uint32_t ra0 = (my_rw & 0xffffu) << 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#if BORNVI_R3_NO_LDS_READ
      asm volatile("" : "=v"(ar[j]), "=v"(ai[j]) : "v"(ra0 ^ comb3(j, RB)));
#else
      const double2 x = *reinterpret_cast<const double2*>(lds + (ra0 ^ comb3(j, RB)));
      ar[j] = x.x; ai[j] = x.y;
#endif
    }
    if (cross) {      // the read map took amplitudes from other threads' groups (plan.hpp: STAGE_CROSS_READ)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
  if (PRE) sign8(my_sg & 0xffu, ar, ai);
  uint32_t wa0 = (IO == 2) ? hbm_off : (my_rw >> 16) << 4;
#if !BORNVI_R3_MAT_UPFRONT
  if (NG > 0) load_u8(reinterpret_cast<const double*>(gb + MAT[0]), Ua);
#endif
  // sig: bit i set = the results of register bit i's gate change places (the record's pivot sits in the other row): slot j
  // of the registers then holds the amplitude of slot j ^ sig -- folded into the write address and the post sign bits
  uint32_t sig = 0;
  if (NG > 1) {
    gate8<0>(ar, ai, Ua);
    sig |= rec_swap(Ua) ? 1u : 0u;
#if !BORNVI_R3_MAT_UPFRONT
    load_u8(reinterpret_cast<const double*>(gb + MAT[1]), Ub);
#endif
    if (NG > 2) load_u8(reinterpret_cast<const double*>(gb + MAT[2]), Ua);
  }
  if (NG > 2) { gate8<1>(ar, ai, Ub); sig |= rec_swap(Ub) ? 2u : 0u; }
  if (NG > 0) sig |= rec_swap(NG == 2 ? Ub : Ua) ? (1u << (NG - 1)) : 0u;
  uint32_t post_bits = my_sg >> 16;
  if (POST && NG > 0) {
    if (sig & 1u) post_bits = ((post_bits & 0x55u) << 1) | ((post_bits >> 1) & 0x55u);
    if (sig & 2u) post_bits = ((post_bits & 0x33u) << 2) | ((post_bits >> 2) & 0x33u);
    if (sig & 4u) post_bits = ((post_bits & 0x0fu) << 4) | ((post_bits >> 4) & 0x0fu);
  }
  if (DOT) {
    if (NG > 0) gate8<(NG > 0 ? NG - 1 : 0)>(ar, ai, NG == 2 ? Ub : Ua);
    double a = acc;
    switch (sig) {      // register slot j holds the amplitude of outcome slot j ^ sig
#define BORNVI_DOT8(S_) case S_: _Pragma("unroll") for (int j = 0; j < 8; ++j) a = fma(fma(ar[j], ar[j], ai[j] * ai[j]), W[j ^ S_], a); break;
      BORNVI_DOT8(0) BORNVI_DOT8(1) BORNVI_DOT8(2) BORNVI_DOT8(3) BORNVI_DOT8(4) BORNVI_DOT8(5) BORNVI_DOT8(6) BORNVI_DOT8(7)
#undef BORNVI_DOT8
      default: break;
    }
    acc = a;
    return;
  }
  if (NG > 0) wa0 ^= (IO == 2) ? comb3_rt(sig, hb) : comb3_rt(sig, WB);
  if (NG > 0) {
    asm volatile("" : "+v"(wa0));     // the write base is formed here, before the last gate
    gate8_last<(NG > 0 ? NG - 1 : 0), POST, IO, FIN>(ar, ai, NG == 2 ? Ub : Ua, post_bits, lds, wa0, WB, hb, hbm_base, scale);
  } else {
    asm volatile("" : "+v"(wa0));
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      double xr = ar[j], xi = ai[j];
      if (POST) { xr = flip_sign(xr, post_bits, j); xi = flip_sign(xi, post_bits, j); }
      put_slot<IO, FIN>(xr, xi, j, lds, wa0, WB, hb, hbm_base, scale);
    }
  }
}

template <int IO, bool FIN, bool DOT = false>
__device__ __forceinline__ void dispatch8(uint32_t kind, char* __restrict__ lds, const char* __restrict__ gb, const uint32_t (&MAT)[3],
                                          uint32_t my_rw, uint32_t my_sg, const uint32_t (&RB)[3], const uint32_t (&WB)[3], d2_t (&v)[8],
                                          uint32_t hbm_off, const uint32_t (&hb)[3], void* hbm_base, bool cross, double scale,
                                          const double (&W)[8], double& acc) {
#define BORNVI_ST8(NG, PRE, POST) \
  case (NG) | ((PRE) << 3) | ((POST) << 4): stage8<NG, PRE, POST, IO, FIN, DOT>(lds, gb, MAT, my_rw, my_sg, RB, WB, v, hbm_off, hb, hbm_base, cross, scale, W, acc); break;
#define BORNVI_ST8_NG(PRE, POST) BORNVI_ST8(0, PRE, POST) BORNVI_ST8(1, PRE, POST) BORNVI_ST8(2, PRE, POST) BORNVI_ST8(3, PRE, POST)
#if BORNVI_R3_NO_GATES
  kind &= ~7u;
#endif
  switch (kind) {
    BORNVI_ST8_NG(0, 0)
    BORNVI_ST8_NG(1, 0)
    BORNVI_ST8_NG(0, 1)
    BORNVI_ST8_NG(1, 1)
    default: break;     // unreachable: build_compact_tables admits only these kinds
  }
#undef BORNVI_ST8_NG
#undef BORNVI_ST8
}

}  // namespace

// DOT (final pass of a multi-pass plan only): instead of the probabilities of circuit b, partials[b * tiles + g] = the
// tile's share of  sum_z wdot[z] q_b(z)  (fixed summation order: deterministic); probs is not written.
template <bool DOT>
__global__ __launch_bounds__(1024) void circuit_pass_r3_kernel(
    const uint32_t* __restrict__ plan, uint32_t pass_off, const uint32_t* __restrict__ ctab, uint32_t ct_off,
    const double2* __restrict__ in, double2* __restrict__ out, double* __restrict__ probs,
    const double* __restrict__ gates, long long gate_stride, long long state_stride, long long total_tiles,
    int direct_mask, PrefixShare share, const double* __restrict__ wdot, double* __restrict__ partials) {
  extern __shared__ double2 tile[];
  const uint32_t* __restrict__ P = plan + pass_off;
  const uint32_t* __restrict__ C = ctab + ct_off;
  const uint32_t flags = P[PW_FLAGS];
  const int k = (int)P[PW_K], n = (int)P[PW_N];
  uint32_t H[CH_WORDS];
#pragma unroll
  for (int i = 0; i < CH_WORDS; ++i) H[i] = C[i];
  // T == max(64, 2^(k-3)) threads; the first 2^(k-3) hold 8 tile elements each (tiles below 2^9: part of one wave)
  const uint32_t t = threadIdx.x, T = blockDim.x;
  const uint32_t lane = t & 63u, wv = t >> 6;
  const int kt = k - 3;
  const uint32_t ksize = 1u << k;
  const int gbits = n - k;
  const int nstages = BORNVI_R3_NO_STAGES ? 0 : (int)H[CH_NSTAGES];
  const uint32_t nrows = H[CH_NROWS], nsign = H[CH_NSIGN], NW = H[CH_NWAVES];
  const uint32_t sign_any = H[CH_SIGN_PRE] | H[CH_SIGN_POST];
  const bool init = flags & PASS_INIT, fin = flags & PASS_FINAL;
  const int out_shift = fin ? 3 : 4;
  // ---- LDS: tile | LANE rows | UNI rows of one tile row | MASK ----
  char* __restrict__ lds = reinterpret_cast<char*>(tile);
  uint32_t* __restrict__ lane_tab = reinterpret_cast<uint32_t*>(tile + ksize);
  uint32_t* __restrict__ uni_tab = lane_tab + nrows * 64u;
  uint32_t* __restrict__ mask_tab = uni_tab + nrows * NW;
  double* __restrict__ red_tab = reinterpret_cast<double*>(lane_tab + (((nrows * 64u + nrows * NW + (nsign ? nsign : 1u) * NW) + 1u) & ~1u));   // [NW] wave sums (DOT)
  // (direct_mask: bit 0 / 1 allow the direct first / last stage; bit 2: walk the tiles backwards; bit 3: support of |0..0>)
  const bool direct_in = (H[CH_DIRECT] & 1u) && !init && nstages > 0 && (direct_mask & 1);
  const bool direct_out = (H[CH_DIRECT] & 2u) && nstages > 1 && (direct_mask & 2);
  const uint32_t in_step[3] = {direct_in ? H[CH_IN_STEP_D] : H[CH_IN_STEP_N], direct_in ? H[CH_IN_STEP_D + 1] : H[CH_IN_STEP_N + 1],
                               direct_in ? H[CH_IN_STEP_D + 2] : H[CH_IN_STEP_N + 2]};
  const uint32_t fill_step[3] = {H[CH_FILL_STEP], H[CH_FILL_STEP + 1], H[CH_FILL_STEP + 2]};
  const uint32_t drain_step[3] = {H[CH_DRAIN_STEP], H[CH_DRAIN_STEP + 1], H[CH_DRAIN_STEP + 2]};
  const uint32_t out_step_d[3] = {H[CH_OUT_STEP_D], H[CH_OUT_STEP_D + 1], H[CH_OUT_STEP_D + 2]};
  const uint32_t out_step_n[3] = {H[CH_OUT_STEP_N], H[CH_OUT_STEP_N + 1], H[CH_OUT_STEP_N + 2]};
  const uint32_t row0 = H[CH_NSTAGES] + nsign;
  const uint32_t row_in = row0 + (direct_in ? CR_IN_D : CR_IN_N);
  const uint32_t row_out_d = row0 + CR_OUT_D, row_out_n = row0 + CR_OUT_N, row_slot = row0 + CR_SLOT;
  const uint32_t* __restrict__ UNI = C + H[CH_UNI_OFF];
  const uint32_t* __restrict__ MASK = C + H[CH_MASK_OFF];
  // tile-row independent tables -> LDS (trip -1 already reads them for the first prefetch)
  for (uint32_t i = t; i < nrows * 64u; i += T) lane_tab[i] = C[H[CH_LANE_OFF] + i];
  __syncthreads();

#if BORNVI_R3_STAGGER
  for (uint32_t q = 0; q < ((blockIdx.x >> 3) & 3u) * (BORNVI_R3_STAGGER / 4); ++q) __builtin_amdgcn_s_sleep(1);
#endif
  double W[8];              // DOT: dL/dq at the thread's 8 outcomes of tile row g_w
  double acc = 0.0;
#pragma unroll
  for (int j = 0; j < 8; ++j) W[j] = 0.0;
  uint32_t g_w = 0xffffffffu;
  d2_t v[8];                // amplitudes of the NEXT tile (in flight during the current tile's stages)
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (d2_t){0.0, 0.0};
  uint32_t g_pref = 0xffffffffu, in_uni = 0;   // tile row whose CR_IN uniform word is in in_uni
  uint32_t g_tab = 0xffffffffu;                // tile row whose UNI / MASK rows are in LDS
  const uint32_t* __restrict__ CS0 = C + CH_WORDS;

#define BORNVI_RUN_STAGE8(S_, IO_)                                                                              \
  do {                                                                                                          \
    const uint32_t* __restrict__ CS_ = CS0 + (S_) * CS_WORDS;                                                   \
    const uint32_t kind_ = CS_[CS_KIND];                                                                        \
    const uint32_t RB_[3] = {CS_[CS_RB], CS_[CS_RB + 1], CS_[CS_RB + 2]};                                       \
    const uint32_t WB_[3] = {CS_[CS_WB], CS_[CS_WB + 1], CS_[CS_WB + 2]};                                       \
    const uint32_t MAT_[3] = {CS_[CS_MAT], CS_[CS_MAT + 1], CS_[CS_MAT + 2]};                                   \
    const uint32_t rw_ = lane_tab[(uint32_t)(S_) * 64u + lane_t] ^ uni_tab[(uint32_t)(S_) * NW + wv_t];             \
    uint32_t sg_ = 0;                                                                                           \
    if (kind_ >> 3) {                                                                                           \
      const uint32_t sr_ = (uint32_t)__popc(sign_any & ((1u << (S_)) - 1u));                                    \
      sg_ = lane_tab[((uint32_t)nstages + sr_) * 64u + lane_t] ^ uni_tab[((uint32_t)nstages + sr_) * NW + wv_t];    \
      const uint32_t mk_ = mask_tab[sr_ * NW + wv_t];                                                             \
      sg_ ^= (0u - ((uint32_t)__popc(lane_t & (mk_ & 0xffu)) & 1u)) & 0x0000ffffu;                                \
      sg_ ^= (0u - ((uint32_t)__popc(lane_t & (mk_ >> 8)) & 1u)) & 0xffff0000u;                                   \
    }                                                                                                           \
    const uint32_t ho_ = (IO_) == 2 ? (lane_tab[row_out_d * 64u + lane_t] ^ uni_tab[row_out_d * NW + wv_t]) : 0u;   \
    if ((IO_) == 2 && fin && DOT)                                                                               \
      dispatch8<IO_, true, DOT>(kind_, lds, gb, MAT_, rw_, sg_, RB_, WB_, v, ho_, out_step_d, hbm_base, CS_[CS_CROSS] != 0u, scale, W, acc); \
    else if ((IO_) == 2 && fin)                                                                                 \
      dispatch8<IO_, true>(kind_, lds, gb, MAT_, rw_, sg_, RB_, WB_, v, ho_, out_step_d, hbm_base, CS_[CS_CROSS] != 0u, scale, W, acc); \
    else                                                                                                        \
      dispatch8<IO_, false>(kind_, lds, gb, MAT_, rw_, sg_, RB_, WB_, v, ho_, out_step_d, hbm_base, CS_[CS_CROSS] != 0u, 1.0, W, acc); \
  } while (0)

  const long long walk_flip = total_tiles - 1;
  const bool walk_rev = (direct_mask & 4) != 0;
  const bool zskip = init && gbits > 0 && total_tiles < (1ll << 31);
  const uint32_t zinfo = (direct_mask & 8) ? H[CH_ZINFO] : 0u;
  const uint32_t zgmask = zskip ? zinfo : 0u;
  const uint32_t zslots = (!init && direct_in) ? (zinfo & 0xffu) : 0u;
  const uint32_t zs_nb = (uint32_t)(total_tiles >> gbits), zs_gm1 = (1u << gbits) - 1u;
  for (long long Scur = (long long)blockIdx.x - (long long)gridDim.x;; Scur += gridDim.x) {
    const bool real = Scur >= 0;
    const long long Snext = Scur + gridDim.x;
    // (per-trip copies of the lane / wave numbers the compiler cannot see through: nothing derived from them -- a dozen
    // LDS addresses of table rows -- is hoisted out of the tile loop and kept in registers across the stages)
    uint32_t lane_t = lane, wv_t = wv;
    asm volatile("" : "+v"(lane_t), "+v"(wv_t));
    const uint32_t t_t = (wv_t << 6) | lane_t;
    // (tiles below 2^9: only the first 2^(k-3) lanes of the one wave hold amplitudes; the fused-dot instantiation is
    // offered for tiles of whole waves only)
    const bool active = DOT ? true : t_t < (ksize >> 3);
    const bool has_next = Snext < total_tiles;
    long long Tcur, Tnext;
    if (zskip) {     // INIT pass: the one non-zero tile of every circuit first, spread over all workgroups, then the zero tiles
      const uint32_t sc = (uint32_t)(real ? Scur : 0), sn = (uint32_t)(has_next ? Snext : 0);
      const uint32_t bc_ = sc < zs_nb ? sc : (sc - zs_nb) / zs_gm1, bn_ = sn < zs_nb ? sn : (sn - zs_nb) / zs_gm1;
      Tcur = sc < zs_nb ? ((long long)sc << gbits) : (((long long)bc_ << gbits) | (1u + (sc - zs_nb - bc_ * zs_gm1)));
      Tnext = sn < zs_nb ? ((long long)sn << gbits) : (((long long)bn_ << gbits) | (1u + (sn - zs_nb - bn_ * zs_gm1)));
    } else {
      Tcur = walk_rev ? walk_flip - Scur : Scur;
      Tnext = walk_rev ? walk_flip - Snext : Snext;
    }
    if (!real && !has_next) break;
    const uint32_t g = real ? (uint32_t)(Tcur & ((1ll << gbits) - 1)) : 0u;
    const long long b = real ? (Tcur >> gbits) : 0;
    const bool zero_tile = zskip && real && g != 0u;
    const bool noop_tile = zero_tile && (g & zgmask) != 0u;
    const char* __restrict__ gb = reinterpret_cast<const char*>(gates + b * gate_stride);   // this circuit's fused matrices
    // (normalised gates: the pivots' |p|^2 multiply back into the probabilities; slot nfused of the circuit's gate array)
    double scale = 1.0;
    if (fin && real) {
      if (!DOT && init) {
        // a single-pass plan (one tile per circuit): the wave multiplies the circuit's |p|^2 itself -- the same lane split
        // and butterfly as gate_scale_kernel (same bits), which the launch-bound sizes then need not launch
        const uint32_t nf_ = plan[PH_NFUSED];
        double sc_ = 1.0;
        for (uint32_t f_ = lane_t; f_ < nf_; f_ += 64u) sc_ *= reinterpret_cast<const double*>(gb)[f_ * 8u + 6u];
#pragma unroll
        for (int off_ = 32; off_ > 0; off_ >>= 1) sc_ *= __shfl_xor(sc_, off_, 64);
        scale = sc_;
      } else {
        scale = *reinterpret_cast<const double*>(gb + (size_t)plan[PH_NFUSED] * 64u);
      }
    }
    double2* dst = out + b * state_stride;
    double* pdst = probs + (b << n);
    if (fin && share.row_map) {
      const int row = share.row_map[b];
      pdst = row < 0 ? share.trash : probs + ((long long)row << n);
    }
    void* hbm_base = fin ? (void*)pdst : (void*)dst;
    if (real) {
      // the tile has arrived in registers: all but this wave's 8 tile-out stores are done (DOT: a trip issues no stores)
      if (DOT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      if (DOT && g != g_w && active) {      // the thread's 8 weights of this tile row (the grid keeps a workgroup on one row: once per launch)
        g_w = g;
        const uint32_t o_ = direct_out ? (lane_tab[row_out_d * 64u + lane_t] ^ uni_tab[row_out_d * NW + wv_t])
                                       : (lane_tab[row_out_n * 64u + lane_t] ^ uni_tab[row_out_n * NW + wv_t]);
#pragma unroll
        for (int j = 0; j < 8; ++j) W[j] = wdot[(o_ ^ (direct_out ? comb3(j, out_step_d) : comb3(j, out_step_n))) >> 3];
      }
      acc = 0.0;
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(v[i]));
      if (init) {
        if (!zero_tile)
          for (uint32_t u = t_t; u < ksize; u += T) tile[u] = make_double2((u == 0 && g == 0) ? 1.0 : 0.0, 0.0);
      } else if (!direct_in && active) {
        const uint32_t slot_t = (lane_tab[row_slot * 64u + lane_t] ^ uni_tab[row_slot * NW + wv_t]) & 0xffffu;
#pragma unroll
        for (int i = 0; i < 8; ++i) tile[slot_t ^ comb3(i, fill_step)] = make_double2(v[i].x, v[i].y);
      }
      if (direct_in && active) BORNVI_RUN_STAGE8(0, 1);
      asm volatile("" ::: "memory");
    }
    // ---- the registers are free: the next tile starts its trip from HBM now (the only load site) ----
    if (has_next) {
      const uint32_t gn_ = (uint32_t)(Tnext & ((1ll << gbits) - 1));
      const long long bn_ = Tnext >> gbits;
      uint32_t tt_ = t_t;
      asm volatile("" : "+v"(tt_));   // (nothing derived from the thread id is hoisted out of the loop)
      if (!init && gn_ != g_pref) {   // rare: compiler-tracked load, waited for inside this branch
        g_pref = gn_;
        in_uni = UNI[((size_t)gn_ * nrows + row_in) * NW + (tt_ >> 6)];
      }
      if (!init && active) {
        const uint32_t base_ = lane_tab[row_in * 64u + (tt_ & 63u)] ^ in_uni;
        const double2* src_ = in + (bn_ >= share.fresh_begin ? 0ll : bn_) * state_stride;
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (!((zslots >> i) & 1u)) async_load16(v[i], base_ ^ comb3(i, in_step), src_);
      }
    }
    if (real) {
      __syncthreads();                          // the tile (or the first stage's result) is in LDS
      for (int s = direct_in ? 1 : 0; s < (zero_tile ? 0 : nstages); ++s) {
        if (s == nstages - 1 && direct_out) {
          if (active) BORNVI_RUN_STAGE8(s, 2);
        } else {
          if (active) BORNVI_RUN_STAGE8(s, 0);
          __syncthreads();
        }
      }
      // ---- tile out: exactly 8 vector-memory stores per wave (the vmcnt waits count them), here or in the last stage ----
      if (noop_tile || !active) {
      } else if (zero_tile) {
        const uint32_t off0 = (xor_cols16(t_t, kt, P + PW_OUT_COL) ^ xor_cols16(g, gbits, P + PW_OUT_GCOL)) << out_shift;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const uint32_t off = off0 ^ comb3(i, out_step_n);
          if (fin) async_store8(off, 0.0, pdst);
          else async_store16(off, (d2_t){0.0, 0.0}, dst);
        }
      } else if (!direct_out) {
        const uint32_t slot_t = (lane_tab[row_slot * 64u + lane_t] ^ uni_tab[row_slot * NW + wv_t]) >> 16;
        const uint32_t off0 = lane_tab[row_out_n * 64u + lane_t] ^ uni_tab[row_out_n * NW + wv_t];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const double2 x = tile[slot_t ^ comb3(i, drain_step)];
          const uint32_t off = off0 ^ comb3(i, out_step_n);
          if (DOT) acc = fma(x.x * x.x + x.y * x.y, W[i], acc);
          else if (fin) async_store8(off, (x.x * x.x + x.y * x.y) * scale, pdst);
          else async_store16(off, (d2_t){x.x, x.y}, dst);
        }
      }
    }
    if (DOT && real) {
      // the tile's share of the dot product: wave shuffles, then the waves' sums in wave order (fixed: deterministic)
      double a_ = acc;
#pragma unroll
      for (int off_ = 32; off_ > 0; off_ >>= 1) a_ += __shfl_xor(a_, off_, 64);
      if (lane_t == 0) red_tab[wv_t] = a_;
      __syncthreads();
      if (t_t == 0) {
        double tot_ = 0.0;
        for (uint32_t w_ = 0; w_ < NW; ++w_) tot_ += red_tab[w_];
        partials[((size_t)b << gbits) + g] = tot_ * scale;
      }
    }
    if (!has_next) break;
    // ---- UNI / MASK rows of the NEXT tile's row -> LDS (the launcher makes the grid a multiple of the tiles per state
    // whenever it can, so this runs once, in trip -1; compiler-tracked loads).  All stages of the current tile are done. ----
    {
      const uint32_t gnx = (uint32_t)(Tnext & ((1ll << gbits) - 1));
      if (gnx != g_tab && !(zskip && gnx != 0u)) {
        g_tab = gnx;
        if (real) __syncthreads();            // (a slower wave may still be reading the rows of the current tile row)
        for (uint32_t i = t_t; i < nrows * NW; i += T) uni_tab[i] = UNI[(size_t)gnx * nrows * NW + i];
        for (uint32_t i = t_t; i < nsign * NW; i += T) mask_tab[i] = MASK[(size_t)gnx * nsign * NW + i];
      }
    }
    // (trip -1 issued no stores: the first real trip's vmcnt(8) would let its 8 loads pass -- wait for them here)
    if (!real) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // the tile is overwritten by the next trip; the table rows are in place
  }
#undef BORNVI_RUN_STAGE8
}

hipError_t prepare_circuit_r3_kernel(size_t lds_bytes) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(circuit_pass_r3_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds_bytes);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(circuit_pass_r3_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)lds_bytes);
}

int circuit_r3_workgroups_per_cu(int threads, size_t lds) {
  int nb = 0, nd = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, circuit_pass_r3_kernel<false>, threads, lds) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nd, circuit_pass_r3_kernel<true>, threads, lds) != hipSuccess) return 0;
  return nb < nd ? nb : nd;
}

hipError_t launch_circuit_pass_r3(const uint32_t* plan, uint32_t pass_off, const uint32_t* ctab, uint32_t ct_off, int n, int k,
                                  size_t lds, int batch, const void* in, void* out, double* probs, const double* gates,
                                  long long gate_stride, int max_workgroups, int direct_mask, const PrefixShare& share,
                                  const double* wdot, double* partials, hipStream_t st) {
  const long long total_tiles = (long long)batch << (n - k);
  if (total_tiles == 0) return hipSuccess;
  long long wgs = (max_workgroups > 0 && total_tiles > max_workgroups) ? max_workgroups : total_tiles;
  const long long per_state = 1ll << (n - k);
  if (wgs > per_state) wgs -= wgs % per_state;      // a workgroup keeps its tile row: its table rows stay in LDS
  if (wdot)
    circuit_pass_r3_kernel<true><<<dim3((unsigned)wgs), dim3(k >= 9 ? 1u << (k - 3) : 64u), lds, st>>>(
        plan, pass_off, ctab, ct_off, (const double2*)in, (double2*)out, probs, gates, gate_stride, 1ll << n, total_tiles,
        direct_mask, share, wdot, partials);
  else
    circuit_pass_r3_kernel<false><<<dim3((unsigned)wgs), dim3(k >= 9 ? 1u << (k - 3) : 64u), lds, st>>>(
        plan, pass_off, ctab, ct_off, (const double2*)in, (double2*)out, probs, gates, gate_stride, 1ll << n, total_tiles,
        direct_mask, share, nullptr, nullptr);
  return hipGetLastError();
}

// ---- gradient from the fused-dot partials (one thread per parameter, tiles summed in order: deterministic) ----
__global__ __launch_bounds__(64) void dot_finish_kernel(const double* __restrict__ partials, int n_shift, long long tiles,
                                                        const double* __restrict__ ksd2, double* __restrict__ grad,
                                                        double* __restrict__ loss_out) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  double scale = 0.5, loss = 0.0;
  if (ksd2) {
    const double k2 = *ksd2;
    loss = sqrt(k2 < 1e-12 ? 1e-12 : k2);
    scale = (k2 < 1e-12) ? 0.0 : 0.5 / loss;
  }
  if (p == 0 && loss_out && ksd2) *loss_out = loss;
  if (p >= n_shift) return;
  const double* pp = partials + (2ll * p) * tiles;
  const double* pm = pp + tiles;
  double sp = 0.0, sm = 0.0;
  for (long long g = 0; g < tiles; ++g) { sp += pp[g]; sm += pm[g]; }
  grad[p] = scale * (sp - sm);
}

hipError_t launch_dot_finish(const double* partials, int n_shift, long long tiles, const double* ksd2, double* grad,
                             double* loss_out, hipStream_t st) {
  const int blocks = n_shift > 0 ? (n_shift + 63) / 64 : 1;
  dot_finish_kernel<<<dim3((unsigned)blocks), dim3(64), 0, st>>>(partials, n_shift, tiles, ksd2, grad, loss_out);
  return hipGetLastError();
}

}  // namespace bornvi
