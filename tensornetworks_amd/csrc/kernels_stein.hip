// Stein-kernel side of the KSD hot path on gfx950: score function from packed CPTs, dense
// Gram matrix K_p, the quadratic form q^T K_p q (HBM-bound GEMV) and the pack / combine steps
// of the matrix-free mat-vec.  All fp64; every reduction has a fixed order (no atomics), so
// results are bitwise reproducible.
//
// Index convention: outcome index i <-> tuple z with z[b] = (i >> (n-1-b)) & 1 (utils.py:77-91).
#include <hip/hip_runtime.h>
#include <cstdlib>

#include <cmath>
#include <type_traits>

#include "kernels.hpp"

namespace bornvi {

// ------------------------------------------------------------------------------------------------
// score: S[z, b] = 1 - p(x, flip_b z) / p(x, z), zeros when |p(x,z)| < 1e-12
// (stein_utils.py:115-136 on top of compute_prob_joint_xz :58-112 and
//  BayesianNetwork.get_joint_probability bayesian_network.py:111-146).  Same operation order as
// the reference: product over nodes in network order, hidden nodes summed in lexicographic order.
// ------------------------------------------------------------------------------------------------
__device__ inline double joint_xz(const bornvi_bn_desc& bn, int n, unsigned long long z, int n_hidden) {
  double tot = 0.0;
  const unsigned long long n_assign = 1ull << n_hidden;
  for (unsigned long long a = 0; a < n_assign; ++a) {
    unsigned long long vals = 0;  // bit v = value of node v
    int h = 0;
    for (int v = 0; v < bn.num_nodes; ++v) {
      const int role = bn.role[v];
      unsigned long long bit;
      if (role >= 0) bit = (z >> (n - 1 - role)) & 1ull;
      else if (role == -1) bit = 0ull;
      else if (role == -2) bit = 1ull;
      else { bit = (a >> (n_hidden - 1 - h)) & 1ull; ++h; }
      vals |= bit << v;
    }
    double prob = 1.0;
    for (int v = 0; v < bn.num_nodes; ++v) {
      int cfg = 0;
      const int np = bn.n_parents[v];
      for (int p = 0; p < np; ++p) cfg = cfg * 2 + (int)((vals >> bn.parents[v * bn.max_parents + p]) & 1ull);
      prob *= bn.cpt[bn.cpt_off[v] + 2 * cfg + (int)((vals >> v) & 1ull)];
    }
    tot += prob;
  }
  return tot;
}

__global__ __launch_bounds__(256) void score_kernel(bornvi_bn_desc bn, int n, double* __restrict__ S, double* __restrict__ pxz) {
  const unsigned long long N = 1ull << n;
  const unsigned long long z = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (z >= N) return;
  int n_hidden = 0;
  for (int v = 0; v < bn.num_nodes; ++v) n_hidden += (bn.role[v] == -3);
  const double p = joint_xz(bn, n, z, n_hidden);
  if (pxz) pxz[z] = p;
  const bool degenerate = fabs(p) < 1e-12;
  for (int b = 0; b < n; ++b) {
    double s = 0.0;
    if (!degenerate) {
      const double pf = joint_xz(bn, n, z ^ (1ull << (n - 1 - b)), n_hidden);
      s = 1.0 - (pf / p);
    }
    S[z * n + b] = s;
  }
}

// ------------------------------------------------------------------------------------------------
// dense Gram.  Closed form of stein_utils.get_stein_kernel_kp_value (:138-197) with the Hamming
// base kernel (:30-55), a = exp(-1/(n l)), d = popcount(i ^ j), T = S - 1:
//   k_p(i,j) = a^d [ sum_b S_ib S_jb - c_same sum_b (T_ib + T_jb) - (c_diff - c_same) sum_{b: i_b != j_b} (T_ib + T_jb) ]
//   c_same = 1 - a, c_diff = 1 - 1/a.
// Every term is evaluated symmetrically in (i, j), so K is bitwise symmetric.
// Thread = two adjacent columns (16-byte stores), workgroup = 512 columns x GRAM_ROWS rows; the
// row-side quantities are wave-uniform.  HBM-write bound (8 * 4^n bytes).
// ------------------------------------------------------------------------------------------------
constexpr int GRAM_ROWS = 64;
struct PowTable { double v[33]; };

template <int NB>
__global__ __launch_bounds__(256) void gram_kernel(const double* __restrict__ S, double* __restrict__ K,
                                                   PowTable apow, double c_same, double dc,
                                                   long long row_begin, long long row_end, long long ld) {
  const long long N = 1ll << NB;
  const long long j0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 2;
  if (j0 >= N) return;
  double Sj[2][NB], Tj[2][NB], RTj[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    RTj[c] = 0.0;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      Sj[c][b] = S[(j0 + c) * NB + b];
      Tj[c][b] = Sj[c][b] - 1.0;
      RTj[c] += Tj[c][b];
    }
  }
  const long long i_begin = row_begin + (long long)blockIdx.y * GRAM_ROWS;
  const long long i_end = (i_begin + GRAM_ROWS < row_end) ? i_begin + GRAM_ROWS : row_end;
  for (long long i = i_begin; i < i_end; ++i) {
    double Si[NB], Ti[NB], RTi = 0.0;
#pragma unroll
    for (int b = 0; b < NB; ++b) { Si[b] = S[i * NB + b]; Ti[b] = Si[b] - 1.0; RTi += Ti[b]; }
    double out[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const unsigned x = (unsigned)(i ^ (j0 + c));
      double dot = 0.0, ms = 0.0;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        dot = fma(Si[b], Sj[c][b], dot);
        const double v = Ti[b] + Tj[c][b];
        ms += ((x >> (NB - 1 - b)) & 1u) ? v : 0.0;
      }
      out[c] = apow.v[__popc(x)] * (dot - c_same * (RTi + RTj[c]) - dc * ms);
    }
    *reinterpret_cast<double2*>(K + (i - row_begin) * ld + j0) = make_double2(out[0], out[1]);
  }
}

template <int NB>
static hipError_t launch_gram_nb(const double* S, double* K, const PowTable& apow, double c_same, double dc,
                                 long long row_begin, long long row_end, long long ld, hipStream_t st) {
  const long long N = 1ll << NB;
  if (row_end <= row_begin) return hipSuccess;
  dim3 grid((unsigned)((N / 2 + 255) / 256), (unsigned)((row_end - row_begin + GRAM_ROWS - 1) / GRAM_ROWS));
  gram_kernel<NB><<<grid, 256, 0, st>>>(S, K, apow, c_same, dc, row_begin, row_end, ld);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// dense Gram on the matrix cores (n >= 8).  The bracket of the closed form above is a rank-(3n+2) bilinear
// form in (i, j): with U_ib = T_ib (1 - 2 i_b) and alpha_i = -c_same sum_b T_ib - dc sum_b i_b T_ib,
//   bracket = sum_b S_ib S_jb  +  sum_b (-dc U_ib) j_b  +  sum_b i_b (-dc U_jb)  +  (alpha_i + alpha_j)
// i.e. three small GEMMs, run as chains of v_mfma_f64_16x16x4_f64 on 16 x 16 tiles.  They are kept in three
// separate accumulators and combined as (acc1 + (acc2 + acc3)) + (alpha_i + alpha_j): acc1 is symmetric term
// by term and acc2(i, j) == acc3(j, i) operation by operation, so K stays BITWISE symmetric (the symmetric
// contraction relies on it).  a^d is applied in the epilogue.  HBM-write bound: 8 * 4^n bytes.
// Workgroup = 4 waves, GM_ROWS rows x GM_COLS columns; the row factors sit in LDS for the whole workgroup, each
// wave walks over 16-column tiles of its share, whose column factors it builds in its own LDS scratch.
// Fragment layout of the f64 MFMA (cdna_hip_programming.md, section 3): A[row = lane & 15][k = lane >> 4],
// B[k = lane >> 4][col = lane & 15], D[row = (lane >> 4) + 4 r][col = lane & 15], r = 0..3.
// ------------------------------------------------------------------------------------------------
constexpr int GM_ROWS = 64, GM_COLS = 2048, GM_MAXN = 17;
typedef double d4_t __attribute__((ext_vector_type(4)));

struct GramFactorPitch {
  int np, p, rowpitch;   // padded K of each of the three products; LDS pitch of one factor row; of one outcome
};
__host__ __device__ constexpr GramFactorPitch gram_pitch(int n) {
  GramFactorPitch g{};
  g.np = (n + 3) / 4 * 4;
  g.p = g.np + 2;              // pitch = 2 (mod 4): the 16 rows x 2 k's a half-wave reads hit 32 distinct 8-byte banks
  g.rowpitch = 3 * g.p + 4;    // F1 | F2 | F3 | alpha, again 2 (mod 4)
  return g;
}

// the n scores of outcome z, all loads issued together (the column side fetches them one tile ahead)
template <int n>
__device__ __forceinline__ void gram_load_scores(const double* __restrict__ S, long long z, double (&sv)[GM_MAXN]) {
#pragma unroll
  for (int b = 0; b < GM_MAXN; ++b) sv[b] = (b < n) ? S[z * n + b] : 0.0;
}

// factors of one outcome index z into its LDS row: F1 = S_z,  F2 = -dc U_z,  F3 = bits(z) (each padded with zeros),
// then alpha_z.  Row side and column side use this same function: alpha_z must be the same bits on both sides.
template <int n>
__device__ __forceinline__ void gram_factors(const double (&sv)[GM_MAXN], long long z, double c_same, double dc,
                                             double* __restrict__ row) {
  constexpr GramFactorPitch g = gram_pitch(n);
  double rt = 0.0, ab = 0.0;
#pragma unroll
  for (int b = 0; b < g.np; ++b) {
    if (b < n) {
      const double t = sv[b] - 1.0;
      const int bit = (int)((z >> (n - 1 - b)) & 1ll);
      rt += t;
      ab += bit ? t : 0.0;
      row[b] = sv[b];
      row[g.p + b] = -dc * (bit ? -t : t);
      row[2 * g.p + b] = bit ? 1.0 : 0.0;
    } else {
      row[b] = 0.0; row[g.p + b] = 0.0; row[2 * g.p + b] = 0.0;
    }
  }
  row[3 * g.p] = -c_same * rt - dc * ab;
}

template <int n>
__global__ __launch_bounds__(256) void gram_mfma_kernel(const double* __restrict__ S, double* __restrict__ K,
                                                        PowTable apow, double c_same, double dc, long long row_begin,
                                                        long long row_end, long long ld) {
  extern __shared__ double gm_lds[];
  __shared__ double apow_s[33];
  constexpr GramFactorPitch g = gram_pitch(n);
  constexpr int KK = g.np / 4;                          // MFMAs per product and tile
  const long long N = 1ll << n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* __restrict__ Lt = gm_lds;                      // [GM_ROWS][rowpitch]
  double* __restrict__ Rt = gm_lds + GM_ROWS * g.rowpitch + wave * 16 * g.rowpitch;   // this wave's [16][rowpitch]
  if (threadIdx.x < 33) apow_s[threadIdx.x] = apow.v[threadIdx.x];
  const long long i_blk = row_begin + (long long)blockIdx.y * GM_ROWS;
  if (threadIdx.x < GM_ROWS) {
    const long long i = i_blk + threadIdx.x;
    double* r = Lt + threadIdx.x * g.rowpitch;
    if (i < row_end) {
      double sv[GM_MAXN];
      gram_load_scores<n>(S, i, sv);
      gram_factors<n>(sv, i, c_same, dc, r);
    } else {
      for (int b = 0; b < g.rowpitch; ++b) r[b] = 0.0;
    }
  }
  __syncthreads();
  const long long j_chunk = (long long)blockIdx.x * GM_COLS;
  const long long j_chunk_end = (j_chunk + GM_COLS < N) ? j_chunk + GM_COLS : N;
  const int ar = lane & 15, ak = lane >> 4;
  double svn[GM_MAXN];     // scores of the NEXT column tile (lanes 0..15), in flight during this tile's MFMAs
  long long j0 = j_chunk + wave * 16;
  if (lane < 16 && j0 < j_chunk_end) gram_load_scores<n>(S, j0 + lane, svn);
  for (; j0 < j_chunk_end; j0 += 64) {
    // column factors of this 16-column tile -> the wave's scratch (lanes 0..15 one column each)
    if (lane < 16) {
      gram_factors<n>(svn, j0 + lane, c_same, dc, Rt + lane * g.rowpitch);
      if (j0 + 64 < j_chunk_end) gram_load_scores<n>(S, j0 + 64 + lane, svn);
    }
    __builtin_amdgcn_wave_barrier();
    // B fragments of the three products (column side), kept in registers across the row tiles
    double b1[KK], b2[KK], b3[KK];
    const double* rj = Rt + ar * g.rowpitch;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      b1[kk] = rj[4 * kk + ak];
      b2[kk] = rj[2 * g.p + 4 * kk + ak];   // pairs with the row side's F2: bits(j)
      b3[kk] = rj[g.p + 4 * kk + ak];       // pairs with the row side's F3: -dc U_j
    }
    const double alpha_j = rj[3 * g.p];
    const long long j = j0 + ar;
#pragma unroll 1
    for (int it = 0; it < GM_ROWS / 16; it += 2) {   // two row tiles at a time: six independent MFMA chains
      const double* ri0 = Lt + (it * 16 + ar) * g.rowpitch;
      const double* ri1 = ri0 + 16 * g.rowpitch;
      d4_t a1 = {0.0, 0.0, 0.0, 0.0}, a2 = a1, a3 = a1, c1 = a1, c2 = a1, c3 = a1;
#ifndef BORNVI_GRAM_NO_MFMA     // (timing-only ablations: tools/probes/build_gram_variants.sh)
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ri0[4 * kk + ak], b1[kk], a1, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ri1[4 * kk + ak], b1[kk], c1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ri0[g.p + 4 * kk + ak], b2[kk], a2, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ri1[g.p + 4 * kk + ak], b2[kk], c2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(ri0[2 * g.p + 4 * kk + ak], b3[kk], a3, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(ri1[2 * g.p + 4 * kk + ak], b3[kk], c3, 0, 0, 0);
      }
#else
      a1[0] = a1[1] = a1[2] = a1[3] = ri0[ak] + b1[0]; c1 = a2 = c2 = a3 = c3 = a1;
#endif
      // D layout: row = (lane >> 4) + 4 r, col = lane & 15
      const long long ib = i_blk + it * 16 + ak;
      const double* al0 = Lt + (it * 16 + ak) * g.rowpitch + 3 * g.p;   // alpha of row ak + 4 r: + 4 r rowpitch
      double* __restrict__ Kp = K + (ib - row_begin) * ld + j;
      if (i_blk + GM_ROWS <= row_end) {     // whole row block inside the range (wave-uniform): no per-row tests
        double w0[4], w1[4], e0[4], e1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          w0[r] = apow_s[__popcll((unsigned long long)((ib + 4 * r) ^ j))];
          w1[r] = apow_s[__popcll((unsigned long long)((ib + 16 + 4 * r) ^ j))];
          e0[r] = al0[4 * r * g.rowpitch] + alpha_j;
          e1[r] = al0[(16 + 4 * r) * g.rowpitch] + alpha_j;
        }
#ifdef BORNVI_GRAM_NO_STORE
        double keep = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          keep += w0[r] * ((a1[r] + (a2[r] + a3[r])) + e0[r]) + w1[r] * ((c1[r] + (c2[r] + c3[r])) + e1[r]);
        if (keep == 1.2345e-300) Kp[0] = keep;
#else
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          __builtin_nontemporal_store(w0[r] * ((a1[r] + (a2[r] + a3[r])) + e0[r]), Kp + (long long)(4 * r) * ld);
          __builtin_nontemporal_store(w1[r] * ((c1[r] + (c2[r] + c3[r])) + e1[r]), Kp + (long long)(16 + 4 * r) * ld);
        }
#endif
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long long i = ib + 4 * r;
          if (i < row_end)
            K[(i - row_begin) * ld + j] = apow_s[__popcll((unsigned long long)(i ^ j))] *
                                         ((a1[r] + (a2[r] + a3[r])) + (al0[4 * r * g.rowpitch] + alpha_j));
          if (i + 16 < row_end)
            K[(i + 16 - row_begin) * ld + j] = apow_s[__popcll((unsigned long long)((i + 16) ^ j))] *
                                              ((c1[r] + (c2[r] + c3[r])) + (al0[(16 + 4 * r) * g.rowpitch] + alpha_j));
        }
      }
    }
    __builtin_amdgcn_wave_barrier();     // the scratch is rewritten by the next tile
  }
}

// ---- the shipped form: ONE matrix product per tile, the two bit products from tables ------------------------------
// Measured on gfx950 (tools/probes/mfma64_probe.hip): v_mfma_f64_16x16x4_f64 holds the SIMD's vector ALU for its 64
// cycles -- no vector instruction of ANY wave on that SIMD issues beside it (6 MFMAs + 96 v_add_u32 on a second wave
// take the sum of their times, not the maximum).  So gram_mfma_kernel's time is its 12 MFMAs (768 cycles per tile)
// PLUS its epilogue (timing-only builds: 6.1 + 3.9 of 10.2 ms at n = 16), software pipelining changes nothing (built,
// measured, bit-identical, 10.1 ms), and the way down is fewer vector-pipe cycles per tile.  The two products with the
// 0/1 matrices need no matrix instruction: with the outcome index split into its tile part T (bits >= 4) and its
// in-tile part c (low 4 bits),
//     X(z, t) := sum_b G_zb t_b = PH(z, T) + PL(z, c),        G_zb = -dc U_zb,
// PH(z, T) is one number per (outcome, 16-wide tile of the other index) and PL(z, c) one of 16 per outcome, and
//     K(i, j) = a^d(i,j) * ( (acc1 + (PL(i, c_j) + PL(j, c_i))) + (A(i, T_j) + A(j, T_i)) ),   A(z, T) = PH(z, T) + alpha_z.
// Every term is either symmetric in (i, j) or one of a pair added commutatively, and PH / PL / alpha are computed by
// the same instruction sequence whichever side z is on: K stays BITWISE symmetric.  Per tile: 4 MFMAs (acc1) and
// ~60 vector instructions instead of 12 and ~75; HBM-write bound from here (8 * 4^n bytes).
// Workgroup = 4 waves x (64 aligned rows) x GM_COLS columns; rows outside [row_begin, row_end) are computed and not stored.
template <int n>
struct GramV3 {
  static constexpr int NP = (n + 3) / 4 * 4;            // scores padded to the MFMA's k = 4
  static constexpr int KK = NP / 4;
  static constexpr int NH = n - 4;                      // tile bits of an outcome index
  static constexpr int OFF_G = NP;                      // row layout in LDS: F1 (MFMA order) | G high | PL table | alpha
  static constexpr int OFF_PL = OFF_G + NH;
  static constexpr int OFF_AL = OFF_PL + 16;
  static constexpr int PITCH = ((OFF_AL + 1 + 3) / 4) * 4 + 2;      // = 2 (mod 4): 16 rows x 16-byte reads spread over the banks
};

// alpha_z and G_zb of one outcome from its scores (the same code on the row side and on the column side)
template <int n>
__device__ __forceinline__ void gram_v3_outcome(const double (&sv)[GM_MAXN], long long z, double c_same, double dc,
                                                double (&G)[GM_MAXN], double& alpha) {
  // (no fused multiply-adds the source does not spell out, here and in the two sums below: a multiplier that is a
  // compile-time 1.0 on one side turns fma(G, 1, acc) into acc + G, and contracting THAT with G's own product would
  // skip a rounding the other side performs -- K would lose its bitwise symmetry)
#pragma clang fp contract(off)
  double rt = 0.0, ab = 0.0;
#pragma unroll
  for (int b = 0; b < n; ++b) {
    const double t = sv[b] - 1.0;
    const bool bit = (z >> (n - 1 - b)) & 1ll;
    rt += t;
    ab += bit ? t : 0.0;
    G[b] = (bit ? dc : -dc) * t;
  }
  alpha = __builtin_fma(-dc, ab, -c_same * rt);      // (spelled out: both sides must round alike)
}

// G and alpha of every outcome, once per build ([2^n][n + 1] doubles: G_z0 .. G_z,n-1, alpha_z): both sides of the
// main kernel read them from here (a wave would otherwise redo them for its 16 columns at every column tile: a third
// of its vector instructions)
template <int n>
__global__ __launch_bounds__(256) void gram_prep_kernel(const double* __restrict__ S, double c_same, double dc,
                                                       double* __restrict__ GA) {
  const long long z = (long long)blockIdx.x * 256 + threadIdx.x;
  if (z >= (1ll << n)) return;
  double sv[GM_MAXN], G[GM_MAXN], alpha;
  gram_load_scores<n>(S, z, sv);
  gram_v3_outcome<n>(sv, z, c_same, dc, G, alpha);
#pragma unroll
  for (int b = 0; b < n; ++b) GA[z * (n + 1) + b] = G[b];
  GA[z * (n + 1) + n] = alpha;
}

template <int n>
__device__ __forceinline__ void gram_load_ga(const double* __restrict__ GA, long long z, double (&G)[GM_MAXN], double& alpha) {
#pragma unroll
  for (int b = 0; b < GM_MAXN; ++b) G[b] = (b < n) ? GA[z * (n + 1) + b] : 0.0;
  alpha = GA[z * (n + 1) + n];
}

// PH(z, T): the tile bits of the other index, most significant first; m = 1.0 / 0.0 per bit (one fma per bit on both sides)
template <int n, int FIRST, int LAST>
__device__ __forceinline__ double gram_v3_ph(const double (&G)[GM_MAXN], long long T, double acc) {
#pragma clang fp contract(off)
  constexpr int NH = n - 4;
#pragma unroll
  for (int b = FIRST; b < LAST; ++b) acc = __builtin_fma(G[b], ((T >> (NH - 1 - b)) & 1ll) ? 1.0 : 0.0, acc);
  return acc;
}

// PL(z, c): the in-tile bits, most significant first
template <int n>
__device__ __forceinline__ double gram_v3_pl(const double (&G)[GM_MAXN], double m3, double m2, double m1, double m0) {
#pragma clang fp contract(off)
  double acc = __builtin_fma(G[n - 4], m3, 0.0);
  acc = __builtin_fma(G[n - 3], m2, acc);
  acc = __builtin_fma(G[n - 2], m1, acc);
  return __builtin_fma(G[n - 1], m0, acc);
}

// the walk of one wave over its column tiles (R rows per workgroup; GUARD: the block sticks out of [row_begin, row_end):
// per-row store tests)
template <int n, int R, bool GUARD>
__device__ __forceinline__ void gram_tables_columns(const double* __restrict__ S, const double* __restrict__ GA,
                                                    const double* __restrict__ Lt_, double* __restrict__ AJ,
                                                    const double* __restrict__ apow_s, int lane, const int (&wslot)[4],
                                                    double mk1, double mk0, unsigned voff, long long i_blk, long long I0,
                                                    double* __restrict__ Kblk, long long ld, long long j0, long long j_chunk_end,
                                                    long long row_begin, long long row_end, int ar, int ak) {
  using L = GramV3<n>;
  constexpr int KK = L::KK, NH = L::NH, PITCH = L::PITCH;
  constexpr int RT = R / 16;                            // row tiles of the block
  constexpr int NV = RT == 8 ? 3 : (RT == 4 ? 2 : 1);   // tile bits that differ between them
  static_assert(RT == 8 || RT == 4, "64 or 128 rows per workgroup");
  // two register sets for the column factors: a tile works on one while the next tile's loads land in the other (a
  // single set rotated through copies cost 44 register moves per column tile)
  double Ga[GM_MAXN], Gb[GM_MAXN], alpha_a, alpha_b, ba[KK], bb[KK];
  auto fetch = [&](long long jt, double (&G)[GM_MAXN], double& alpha, double (&bf)[KK]) {
    gram_load_ga<n>(GA, jt + ar, G, alpha);
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) bf[kk] = (4 * kk + ak < n) ? S[(jt + ar) * n + 4 * kk + ak] : 0.0;
  };
  auto tile_column = [&](long long jt, const double (&G)[GM_MAXN], double alpha_j, const double (&b1)[KK]) {
    const long long J = jt >> 4;
    // (the row block's LDS image through an offset the compiler cannot see through: its loop-invariant reads -- 16 to 32 A
    // fragments, as many PL values -- are re-read per column tile instead of being hoisted into 100+ registers)
    int opaque0 = 0;
    asm volatile("" : "+v"(opaque0));
    const double* __restrict__ Lt = Lt_ + opaque0;
    // column side: A(j, I) for the block's row tiles (they share all but the last NV tile bits), PL(j, c_i)
    const double ph_hi = gram_v3_ph<n, 0, NH - NV>(G, I0, 0.0);
    double PLc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) PLc[r] = gram_v3_pl<n>(G, (r & 2) ? 1.0 : 0.0, (r & 1) ? 1.0 : 0.0, mk1, mk0);
    // row side: A(row, J) of the lane's own rows (lane, lane + 64, ...: their G from LDS), handed to the lanes that need
    // them through the wave's scratch
#pragma unroll
    for (int h = 0; h < R / 64; ++h) {
      const int row = lane + 64 * h;
      const double* gr = Lt + row * PITCH + L::OFF_G;
      double Gr[GM_MAXN];
#pragma unroll
      for (int b = 0; b < GM_MAXN; ++b) Gr[b] = (b < NH) ? gr[b] : 0.0;
      AJ[(row & ~15) | ((row & 3) << 2) | ((row >> 2) & 3)] = gram_v3_ph<n, 0, NH>(Gr, J, 0.0) + gr[L::OFF_AL - L::OFF_G];
    }
    __builtin_amdgcn_wave_barrier();
    double* __restrict__ Kc = Kblk + jt;
    // (one row tile per trip, not unrolled: eight tiles' worth of hoisted LDS reads do not fit the 256 registers of two waves
    // per SIMD, and a tile's MFMAs cannot overlap another tile's vector work anyway -- same pipe)
#pragma unroll 1
    for (int rt = 0; rt < RT; ++rt) {
      const double Bj_rt = gram_v3_ph<n, NH - NV, NH>(G, I0 + rt, ph_hi) + alpha_j;
      const double* ri = Lt + (rt * 16 + ar) * PITCH + ak * KK;
      d4_t a1 = {0.0, 0.0, 0.0, 0.0};
#ifndef BORNVI_GRAM_NO_MFMA
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ri[kk], b1[kk], a1, 0, 0, 0);
#else
      a1[0] = a1[1] = a1[2] = a1[3] = ri[0] + b1[0];
#endif
      const double* aj = AJ + rt * 16 + ak * 4;
      const double* plr = Lt + (rt * 16 + ak) * PITCH + L::OFF_PL + ar;      // PL(row ak + 4 r of the tile, c = ar)
      const double* wt = apow_s + __popcll((unsigned long long)((I0 + rt) ^ J));
      double* __restrict__ rowp = Kc + (long long)(rt * 16) * ld;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double val = wt[wslot[r]] * ((a1[r] + (plr[4 * r * PITCH] + PLc[r])) + (aj[r] + Bj_rt));
        const long long i = i_blk + rt * 16 + ak + 4 * r;
#ifdef BORNVI_GRAM_NO_STORE      // (timing-only ablation: tools/probes/build_gram_variants.sh)
        if (val == 1.2345e-300)
#else
        if (!GUARD || (i >= row_begin && i < row_end))
#endif
#ifdef BORNVI_GRAM_PLAIN_STORE
          *reinterpret_cast<double*>(reinterpret_cast<char*>(rowp + (long long)(4 * r) * ld) + voff) = val;
#else
          __builtin_nontemporal_store(val, reinterpret_cast<double*>(reinterpret_cast<char*>(rowp + (long long)(4 * r) * ld) + voff));
#endif
      }
    }
    __builtin_amdgcn_wave_barrier();     // the scratch is rewritten by the next column tile
  };
  fetch(j0, Ga, alpha_a, ba);
#pragma unroll 1
  for (;;) {
    if (j0 + 64 < j_chunk_end) fetch(j0 + 64, Gb, alpha_b, bb);
    tile_column(j0, Ga, alpha_a, ba);
    j0 += 64;
    if (j0 >= j_chunk_end) break;
    if (j0 + 64 < j_chunk_end) fetch(j0 + 64, Ga, alpha_a, ba);
    tile_column(j0, Gb, alpha_b, bb);
    j0 += 64;
    if (j0 >= j_chunk_end) break;
  }
}

template <int n, int R>
__global__ __launch_bounds__(256, 2) void gram_tables_kernel(const double* __restrict__ S, double* __restrict__ K,
                                                            const double* __restrict__ GA, PowTable apow, double c_same,
                                                            double dc, long long row_begin, long long row_end, long long ld) {
  using L = GramV3<n>;
  extern __shared__ double gm_lds[];
  __shared__ double apow_s[40];
  constexpr int KK = L::KK, NH = L::NH, PITCH = L::PITCH;
  const long long N = 1ll << n;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  double* __restrict__ Lt = gm_lds;                        // [R][PITCH]
  double* __restrict__ AJ = gm_lds + R * PITCH + wave * R;   // this wave's A(row, J), in [row tile][ak][r] order
  if (threadIdx.x < 40) apow_s[threadIdx.x] = threadIdx.x < 33 ? apow.v[threadIdx.x] : 0.0;
  const long long i_blk = (row_begin & ~(long long)(R - 1)) + (long long)blockIdx.y * R;   // R-aligned
  if (threadIdx.x < R) {
    const long long i = i_blk + threadIdx.x;
    double sv[GM_MAXN], G[GM_MAXN], alpha;
    gram_load_scores<n>(S, i, sv);
    gram_load_ga<n>(GA, i, G, alpha);
    double* r = Lt + threadIdx.x * PITCH;
#pragma unroll
    for (int b = 0; b < L::NP; ++b) r[(b & 3) * KK + (b >> 2)] = (b < n) ? sv[b] : 0.0;    // lane ak reads its KK values in one go
#pragma unroll
    for (int b = 0; b < NH; ++b) r[L::OFF_G + b] = G[b];
#pragma unroll
    for (int c = 0; c < 16; ++c)
      r[L::OFF_PL + c] = gram_v3_pl<n>(G, (c & 8) ? 1.0 : 0.0, (c & 4) ? 1.0 : 0.0, (c & 2) ? 1.0 : 0.0, (c & 1) ? 1.0 : 0.0);
    r[L::OFF_AL] = alpha;
  }
  __syncthreads();
  const long long j_chunk = (long long)blockIdx.x * GM_COLS;
  const long long j_chunk_end = (j_chunk + GM_COLS < N) ? j_chunk + GM_COLS : N;
  long long j0 = j_chunk + wave * 16;
  if (j0 >= j_chunk_end) return;
  const int ar = lane & 15, ak = lane >> 4;
  int wslot[4];                                         // a^d table: popcount of the in-tile bits of i ^ j
#pragma unroll
  for (int r = 0; r < 4; ++r) wslot[r] = __popc((unsigned)((ak + 4 * r) ^ ar));
  const double mk1 = (ak & 2) ? 1.0 : 0.0, mk0 = (ak & 1) ? 1.0 : 0.0;     // low bits of this lane's rows ak + 4 r
  const unsigned voff = (unsigned)(((long long)ak * ld + ar) * 8);
  const bool whole = i_blk >= row_begin && i_blk + R <= row_end;            // (uniform) no per-row tests
  const long long I0 = i_blk >> 4;                     // tile index of row tile 0
  double* __restrict__ Kblk = K + (i_blk - row_begin) * ld;                 // (not dereferenced for rows outside the range)
  if (whole)
    gram_tables_columns<n, R, false>(S, GA, Lt, AJ, apow_s, lane, wslot, mk1, mk0, voff, i_blk, I0, Kblk, ld, j0, j_chunk_end,
                                     row_begin, row_end, ar, ak);
  else
    gram_tables_columns<n, R, true>(S, GA, Lt, AJ, apow_s, lane, wslot, mk1, mk0, voff, i_blk, I0, Kblk, ld, j0, j_chunk_end,
                                    row_begin, row_end, ar, ak);
}

template <int NB>
static hipError_t launch_gram_mfma_nb(const double* S, double* K, const PowTable& apow, double c_same, double dc,
                                      long long row_begin, long long row_end, long long ld, hipStream_t st) {
  const long long N = 1ll << NB;
  if (row_end <= row_begin) return hipSuccess;
  constexpr GramFactorPitch g = gram_pitch(NB);
  const size_t lds = (size_t)(GM_ROWS + 4 * 16) * g.rowpitch * sizeof(double);
  {   // (per device and cheap: set on every launch rather than behind a process-wide flag)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gram_mfma_kernel<NB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  const long long cols = N < GM_COLS ? N : GM_COLS;
  // BORNVI_GRAM_TABLES=0: the three-product kernel (A/B; one process builds every matrix with one of the two -- their
  // roundings differ, and K's two triangles must come from the same arithmetic)
  static const bool tables = [] { const char* e = getenv("BORNVI_GRAM_TABLES"); return !(e && e[0] == '0'); }();
  if (tables) {
    // rows per workgroup: 64.  128 (BORNVI_GRAM_ROWS=128, A/B) halves the column side's work per tile and measures the
    // same at n = 16 (6.9 ms both: with the tables the kernel waits for its stores -- 6.9 ms without the MFMAs too, 6.1 ms
    // without the stores) and slower for small matrices (n = 9: 47 against 37 us)
    static const bool rows64 = [] { const char* e = getenv("BORNVI_GRAM_ROWS"); return !(e && e[0] == '1'); }();
    double* GA = nullptr;                 // stream-ordered scratch: (n + 1) 2^n doubles (8.9 MB at n = 16)
    hipError_t e = hipMallocAsync((void**)&GA, (size_t)(NB + 1) * (size_t)N * sizeof(double), st);
    if (e != hipSuccess) return e;
    gram_prep_kernel<NB><<<(unsigned)((N + 255) / 256), 256, 0, st>>>(S, c_same, dc, GA);
    auto run = [&](auto rows_tag) -> hipError_t {
      constexpr int R = decltype(rows_tag)::value;
      const size_t lds3 = (size_t)(R * GramV3<NB>::PITCH + 4 * R) * sizeof(double);
      hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(gram_tables_kernel<NB, R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
      if (e1 != hipSuccess) return e1;
      const long long first = row_begin & ~(long long)(R - 1);
      dim3 grid((unsigned)((N + cols - 1) / cols), (unsigned)((row_end - first + R - 1) / R));
      gram_tables_kernel<NB, R><<<grid, 256, lds3, st>>>(S, K, GA, apow, c_same, dc, row_begin, row_end, ld);
      return hipGetLastError();
    };
    e = (rows64 || NB < 9) ? run(std::integral_constant<int, 64>{}) : run(std::integral_constant<int, 128>{});
    hipError_t e2 = hipFreeAsync(GA, st);
    return e != hipSuccess ? e : e2;
  }
  dim3 grid((unsigned)((N + cols - 1) / cols), (unsigned)((row_end - row_begin + GM_ROWS - 1) / GM_ROWS));
  gram_mfma_kernel<NB><<<grid, 256, lds, st>>>(S, K, apow, c_same, dc, row_begin, row_end, ld);
  return hipGetLastError();
}

static hipError_t launch_gram_mfma(int n, const double* S, double* K, const PowTable& apow, double c_same, double dc,
                                   long long row_begin, long long row_end, long long ld, hipStream_t st) {
  switch (n) {
#define BORNVI_GM_CASE(NB) case NB: return launch_gram_mfma_nb<NB>(S, K, apow, c_same, dc, row_begin, row_end, ld, st);
    BORNVI_GM_CASE(8) BORNVI_GM_CASE(9) BORNVI_GM_CASE(10) BORNVI_GM_CASE(11) BORNVI_GM_CASE(12) BORNVI_GM_CASE(13)
    BORNVI_GM_CASE(14) BORNVI_GM_CASE(15) BORNVI_GM_CASE(16) BORNVI_GM_CASE(17)
#undef BORNVI_GM_CASE
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_gram_build(int n, double length_scale, const double* S, double* K, long long row_begin,
                             long long row_end, long long ld, hipStream_t st) {
  PowTable apow;
  const double denom = (double)n * length_scale;
  for (int d = 0; d <= 32; ++d) apow.v[d] = std::exp(-(double)d / denom);  // stein_utils.py:55
  const double a = apow.v[1];
  const double c_same = 1.0 - a, c_diff = 1.0 - 1.0 / a;
  const double dc = c_diff - c_same;
  // matrix-core path for n >= 8 (2^n >= 256: whole 16 x 16 tiles); BORNVI_GRAM_VALU=1 keeps the VALU kernel (A/B)
  static const bool force_valu = [] { const char* e = getenv("BORNVI_GRAM_VALU"); return e && e[0] == '1'; }();
  if (n >= 8 && n <= GM_MAXN && !force_valu) return launch_gram_mfma(n, S, K, apow, c_same, dc, row_begin, row_end, ld, st);
  switch (n) {
#define BORNVI_GRAM_CASE(NB) case NB: return launch_gram_nb<NB>(S, K, apow, c_same, dc, row_begin, row_end, ld, st);
    BORNVI_GRAM_CASE(1) BORNVI_GRAM_CASE(2) BORNVI_GRAM_CASE(3) BORNVI_GRAM_CASE(4) BORNVI_GRAM_CASE(5)
    BORNVI_GRAM_CASE(6) BORNVI_GRAM_CASE(7) BORNVI_GRAM_CASE(8) BORNVI_GRAM_CASE(9) BORNVI_GRAM_CASE(10)
    BORNVI_GRAM_CASE(11) BORNVI_GRAM_CASE(12) BORNVI_GRAM_CASE(13) BORNVI_GRAM_CASE(14) BORNVI_GRAM_CASE(15)
    BORNVI_GRAM_CASE(16) BORNVI_GRAM_CASE(17)
#undef BORNVI_GRAM_CASE
    default: return hipErrorInvalidValue;
  }
}

// k_p for M explicit pairs (the batched form of one get_stein_kernel_kp_value call, stein_utils.py:138-197):
// zi/zj outcome indices, si/sj [M, n] score rows supplied by the caller.  Same closed form as gram_kernel.
__global__ __launch_bounds__(256) void kp_pairs_kernel(int n, PowTable apow, double c_same, double dc, long long M,
                                                       const long long* __restrict__ zi, const long long* __restrict__ zj,
                                                       const double* __restrict__ si, const double* __restrict__ sj,
                                                       double* __restrict__ out) {
  const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const unsigned x = (unsigned)(zi[m] ^ zj[m]);
  double dot = 0.0, ms = 0.0, rti = 0.0, rtj = 0.0;
  for (int b = 0; b < n; ++b) {
    const double a_ = si[m * n + b], b_ = sj[m * n + b];
    const double ta = a_ - 1.0, tb = b_ - 1.0;
    dot = fma(a_, b_, dot);
    rti += ta; rtj += tb;
    ms += ((x >> (n - 1 - b)) & 1u) ? (ta + tb) : 0.0;
  }
  out[m] = apow.v[__popc(x)] * (dot - c_same * (rti + rtj) - dc * ms);
}

hipError_t launch_kp_pairs(int n, double length_scale, long long M, const long long* zi, const long long* zj,
                           const double* si, const double* sj, double* out, hipStream_t st) {
  if (M <= 0) return hipSuccess;
  PowTable apow;
  const double denom = (double)n * length_scale;
  for (int d = 0; d <= 32; ++d) apow.v[d] = std::exp(-(double)d / denom);
  const double a = apow.v[1];
  const double c_same = 1.0 - a, c_diff = 1.0 - 1.0 / a;
  kp_pairs_kernel<<<(unsigned)((M + 255) / 256), 256, 0, st>>>(n, apow, c_same, c_diff - c_same, M, zi, zj, si, sj, out);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// quadratic form: y = K q (row-major K, one pass over 8 * 4^n bytes), ksd2 = q . y.
// One wave owns QF_ROWS consecutive rows and streams them with 16-byte non-temporal loads; q stays
// in L2.  Wave reduction by xor-shuffles, then per-workgroup partials, then one summing kernel.
// ------------------------------------------------------------------------------------------------
constexpr int QF_ROWS = 4;
constexpr int QF_WAVES = 4;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ __launch_bounds__(64 * QF_WAVES) void quadform_kernel(const double* __restrict__ K, const double* __restrict__ q,
                                                                 double* __restrict__ y, double* __restrict__ partials, long long N,
                                                                 long long row_begin, long long NR) {
  __shared__ double wpart[QF_WAVES];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long row0 = ((long long)blockIdx.x * QF_WAVES + wave) * QF_ROWS;
  double acc[QF_ROWS];
#pragma unroll
  for (int r = 0; r < QF_ROWS; ++r) acc[r] = 0.0;
  if (row0 < NR) {
    const double* __restrict__ Kr = K + row0 * N;
    if (row0 + QF_ROWS <= NR) {
#pragma unroll 4
      for (long long c = lane * 2; c < N; c += 128) {
        const double2 q2 = *reinterpret_cast<const double2*>(q + c);
#pragma unroll
        for (int r = 0; r < QF_ROWS; ++r) {
          const double* p = Kr + r * N + c;
          const double kx = __builtin_nontemporal_load(p), ky = __builtin_nontemporal_load(p + 1);
          acc[r] = fma(kx, q2.x, fma(ky, q2.y, acc[r]));
        }
      }
    } else {
      for (long long c = lane * 2; c < N; c += 128) {
        const double2 q2 = *reinterpret_cast<const double2*>(q + c);
        for (int r = 0; r < QF_ROWS; ++r)
          if (row0 + r < NR) {
            const double* p = Kr + r * N + c;
            acc[r] = fma(p[0], q2.x, fma(p[1], q2.y, acc[r]));
          }
      }
    }
  }
  double part = 0.0;
#pragma unroll
  for (int r = 0; r < QF_ROWS; ++r) {
    acc[r] = wave_sum(acc[r]);
    if (row0 + r < NR) {
      if (lane == 0 && y) y[row0 + r] = acc[r];
      part += q[row_begin + row0 + r] * acc[r];
    }
  }
  if (lane == 0) wpart[wave] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < QF_WAVES; ++w) s += wpart[w];
    partials[blockIdx.x] = s;
  }
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const double* __restrict__ partials, long long count, double* __restrict__ out) {
  __shared__ double red[256];
  double acc = 0.0;
  for (long long i = threadIdx.x; i < count; i += 256) acc += partials[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = red[0];
}

size_t quadform_partials(long long rows) {
  return (size_t)((rows + QF_ROWS * QF_WAVES - 1) / (QF_ROWS * QF_WAVES));
}

// K holds rows [row_begin, row_end) of the Gram matrix; y (if given) receives those rows of K q,
// ksd2 the partial sum over them of q_i y_i.
hipError_t launch_quadform(int n, const double* K, long long row_begin, long long row_end, const double* q,
                           double* y_or_null, double* ksd2, double* partials, hipStream_t st) {
  const long long N = 1ll << n;
  const long long NR = row_end - row_begin;
  const size_t nwg = quadform_partials(NR);
  if (nwg == 0) return hipMemsetAsync(ksd2, 0, sizeof(double), st);
  quadform_kernel<<<(unsigned)nwg, 64 * QF_WAVES, 0, st>>>(K, q, y_or_null, partials, N, row_begin, NR);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  sum_partials_kernel<<<1, 256, 0, st>>>(partials, (long long)nwg, ksd2);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// symmetric quadratic form: K_p is symmetric (gram_kernel evaluates every entry symmetrically, so it
// is BITWISE symmetric), hence y = K q needs only the upper triangle: half the HBM traffic.
//
// Work split.  The rows are cut into BANDS of SYM_BAND = 256 rows; a workgroup of eight waves owns a band (paired
// with its mirror band for balance) and one piece of its columns; wave w of the workgroup owns the 32 rows
// [i0 + 32 w, i0 + 32 w + 32) of the band and sweeps the columns j >= its first row:
//   row part     yrow[i] = sum_j K_ij q_j over the wave's columns   (32 accumulators per lane, wave-reduced at the end)
//   column part  z[j]    = sum_{i in the wave's rows} K_ij q_i       (2 per lane and 128-column chunk)
// The column partials have to leave the wave once per chunk.  Round 1 stored them per 32-row strip (N^2 / 64 doubles
// = 0.54 GB written and re-read at n = 16, 3 % of the reads) -- and those stores, mixed into the read stream, cost the
// kernel 8 % (2.85 vs 2.63 ms without them) and made its time depend on where the matrix had been allocated (2.85 ..
// 3.35 ms: read/write turn-arounds on whichever HBM channels the two buffers share; measured with timing-only builds,
// tools/probes/sym_probe.py -- neither the row pitch, the TLB reach nor the number of loads in flight mattered).
// Here the eight waves of a band add their column partials through LDS, in fixed order, and ONE 4 KiB store per
// 512-column trip leaves the workgroup: N^2 / 512 doubles (0.07 GB).  Only the columns next to the diagonal, where
// the waves' ranges differ, keep per-wave partials (< 512 doubles per wave).  The second kernel adds, for each
// column j, the row result and every partial above it: no atomics, fixed summation order, deterministic.
//
// Layout facts used below: band S covers rows [256 S, 256 S + 256); its BULK are the columns from
// bulk_start(S) = min(N, 512 (S / 2 + 1)) on -- right of the diagonal blocks of all its waves and a multiple of the
// 512-column trip; the NEAR part of wave w are the columns [128-aligned start of its diagonal block, bulk_start).
// ------------------------------------------------------------------------------------------------
#ifndef BORNVI_SYM_ABLATE      // timing-only builds (tools/probes): 2 = no column part at all (results invalid)
#define BORNVI_SYM_ABLATE 0
#endif
#ifndef BORNVI_SYM_STORE_NT
#define BORNVI_SYM_STORE_NT 0      // 1: the column-partial stores non-temporal (A/B)
#endif
#ifndef BORNVI_SYM_DEFER_STORE
#define BORNVI_SYM_DEFER_STORE 1   // 0: store a trip's column partial at once (A/B)
#endif
#ifndef BORNVI_SYM_WAVES
#define BORNVI_SYM_WAVES 8     // measured on six 32 GiB allocations held at once: 2.66-2.82 ms (8 waves) vs 2.66-2.94 ms (4)
#endif
constexpr int SYM_ROWS = 32;                        // rows per wave (its row accumulators)
constexpr int SYM_WAVES = BORNVI_SYM_WAVES;         // 4 or 8 waves per band
constexpr int SYM_BAND = SYM_ROWS * SYM_WAVES;      // rows per workgroup: the unit of the strip-pair shard
constexpr int SYM_BP = 512 / SYM_BAND;              // bands per 512-column trip width
constexpr int SYM_NEAR = 512;                       // per-wave capacity of near-diagonal column partials (< 480 used)
constexpr int SYM_MAX_PARTS = 16;                   // column pieces per band (row-partial buffers in the workspace)
constexpr int SYM_MIN_N = (SYM_WAVES == 4) ? 8 : 9; // smaller matrices (< 2 bands) go through the full-matrix kernel
// A dense matrix below this size also takes the full-matrix kernel from the single-GPU entry point: with 2^n / 256 bands
// there are too few workgroups to fill the chip, and the whole matrix is cache-sized.  Measured (tools/probes/
// sym_vs_full_probe.py, band / full): n = 9: 41 / 24 us, 10: 69 / 24, 11: 90 / 24, 12: 93 / 33, 13: 110 / 86,
// 14: 225 / 344, 15: 676 / 1270.  (The strip-pair shard of several GPUs keeps the band kernel.)
constexpr int SYM_FULL_BELOW_N = 14;
typedef double sym_d2 __attribute__((ext_vector_type(2)));
typedef unsigned int sym_u4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline long long sym_bulk_start(long long S, long long N) {
  const long long b = 512 * (S / SYM_BP + 1);
  return b < N ? b : N;
}
// offset (doubles) of band S's bulk partials Z2_S[j - bulk_start(S)], j in [bulk_start(S), N): bands in order
__host__ __device__ inline long long sym_z2_offset(long long S, long long N) {
  if (N <= 512) return 0;
  const long long m = S / SYM_BP, r = S % SYM_BP;
  return SYM_BP * m * N - 256 * SYM_BP * m * (m + 1) + r * (N - 512 * (m + 1));
}

// 16 bytes through the wave's buffer descriptor (its 32 rows): address = base + voff (per lane) + soff (wave-uniform:
// row and column position, formed by the scalar unit).  With 64-bit row pointers the 32 pointers and the 32 q_i of a
// wave together exceed its 102 SGPRs (the round-1 kernel carried 374 SGPR spills).  aux = 2: nt (read once).
__device__ __forceinline__ sym_d2 sym_load16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  const sym_u4 w = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 2);
  sym_d2 o;
  o.x = __hiloint2double((int)w.y, (int)w.x);
  o.y = __hiloint2double((int)w.w, (int)w.z);
  return o;
}

// One trip of a wave over CH 128-column chunks starting at column cb: 32 rows x CH chunks, RB rows per batch of loads.
// acc[r] += sum_j K_rj q4_j (row part); z[c] = sum_r K_rj q_r (column part of this trip).
// PEND: the column partial of the PREVIOUS trip (`pend` -> *pend_ptr) is stored here, BEHIND this trip's first batch of
// loads.  On CDNA the vmcnt counter retires loads and stores in issue order: a store issued between two trips sits in
// front of the next trip's loads, and the first s_waitcnt for those loads also waits for the store's acknowledgement --
// one exposed write latency per trip (microseconds while the read stream saturates the memory system, and dependent on
// which channels the workspace and K_p share: that was the "placement" effect, tools/probes/ws_place_probe.py).  Behind
// the first batch the store has the rest of the trip to complete before anything waits on it.
template <int CH, int RB, bool PEND = false>
__device__ __forceinline__ void sym_trip(__amdgpu_buffer_rsrc_t rsrc, long long ld, unsigned vlane, long long cb,
                                         const double2 (&q4)[CH], double (&acc)[SYM_ROWS], const double (&qi)[SYM_ROWS],
                                         double (&z)[CH][2], double pend = 0.0, double* pend_ptr = nullptr) {
  unsigned ld8 = (unsigned)(ld * 8);
  asm volatile("" : "+s"(ld8));      // row offsets are formed per trip by the scalar unit (hoisted they spill: 32 x CH values)
  unsigned soff[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) { z[c][0] = 0.0; z[c][1] = 0.0; soff[c] = (unsigned)((cb + c * 128) * 8); }
#pragma unroll
  for (int r0 = 0; r0 < SYM_ROWS; r0 += RB) {
    sym_d2 kv[RB][CH];
#pragma unroll
    for (int u = 0; u < RB; ++u)
#pragma unroll
      for (int c = 0; c < CH; ++c) kv[u][c] = sym_load16(rsrc, vlane, soff[c] + (unsigned)(r0 + u) * ld8);
    if (PEND && r0 == 0) {
#if BORNVI_SYM_STORE_NT
      __builtin_nontemporal_store(pend, pend_ptr);
#else
      *pend_ptr = pend;
#endif
    }
#pragma unroll
    for (int u = 0; u < RB; ++u)
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        acc[r0 + u] = fma(kv[u][c].x, q4[c].x, fma(kv[u][c].y, q4[c].y, acc[r0 + u]));
        if (BORNVI_SYM_ABLATE < 2) {
          z[c][0] = fma(kv[u][c].x, qi[r0 + u], z[c][0]);
          z[c][1] = fma(kv[u][c].y, qi[r0 + u], z[c][1]);
        }
      }
  }
}

// One band for one workgroup: wave `wave` owns rows [SYM_BAND S + 32 wave, + 32).  part 0 also sweeps the NEAR columns.
// All waves run the same number of bulk trips (one workgroup barrier each).
__device__ __forceinline__ void quadform_sym_band(const double* __restrict__ Kb /* first row of the band */, long long ld,
                                                  const double* __restrict__ q, double* __restrict__ yrow,
                                                  double* __restrict__ Z1, double* __restrict__ Z2, long long N, long long S,
                                                  int lane, int wave, int part, int nparts, sym_d2 (*zbuf)[SYM_WAVES][256]) {
  const long long i0 = S * SYM_BAND + (long long)wave * SYM_ROWS;        // first row of this wave
  const double* __restrict__ Kr = Kb + (long long)wave * SYM_ROWS * ld;
  const unsigned vlane = (unsigned)lane * 16u;
  double qi[SYM_ROWS], acc[SYM_ROWS];
#pragma unroll
  for (int r = 0; r < SYM_ROWS; ++r) { qi[r] = q[i0 + r]; acc[r] = 0.0; }
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(Kr), 0, (int)((long long)SYM_ROWS * ld * 8), 0x00020000);
  const long long bulk0 = sym_bulk_start(S, N);
  if (part == 0) {
    // NEAR columns, one 128-column chunk at a time: lanes left of the wave's first row (lower triangle) contribute
    // nothing, lanes inside its diagonal block contribute to the row part only; per-wave column partials Z1
    double* __restrict__ Z1s = Z1 + (S * SYM_WAVES + wave) * SYM_NEAR - (i0 + SYM_ROWS);     // Z1s[j]
#pragma unroll 1
    for (long long cb = (i0 / 128) * 128; cb < bulk0; cb += 128) {
      const long long col = cb + lane * 2;
      double2 q4[1];
      q4[0] = (col >= i0) ? *reinterpret_cast<const double2*>(q + col) : make_double2(0.0, 0.0);   // (i0 is even)
      double z[1][2];
      sym_trip<1, 8>(rsrc, ld, vlane, cb, q4, acc, qi, z);
      if (BORNVI_SYM_ABLATE < 2 && col >= i0 + SYM_ROWS) *reinterpret_cast<double2*>(Z1s + col) = make_double2(z[0][0], z[0][1]);
    }
  }
  // BULK: trips of 512 columns; the four waves' column partials are added through LDS (fixed order) and stored once
  const long long ntrips = (N - bulk0) / 512;
  const long long tp = (ntrips + nparts - 1) / nparts;
  const long long t0 = (part * tp < ntrips) ? part * tp : ntrips;
  const long long t1 = (t0 + tp < ntrips) ? t0 + tp : ntrips;
  double* __restrict__ Z2s = Z2 + sym_z2_offset(S, N) - bulk0;            // Z2s[j]
  constexpr bool DEFER = SYM_WAVES == 8 && BORNVI_SYM_ABLATE < 2 && BORNVI_SYM_DEFER_STORE;
  // (deferred store, see sym_trip: the first trip "stores" 0.0 to the address its own result goes to one trip later)
  double pend = 0.0;
  double* pend_ptr = Z2s + (bulk0 + t0 * 512) + (wave * 64 + lane);
#pragma unroll 1
  for (long long t = t0; t < t1; ++t) {
    const long long cb = bulk0 + t * 512;
    double2 q4[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) q4[c] = *reinterpret_cast<const double2*>(q + cb + c * 128 + lane * 2);
    double z[4][2];
    sym_trip<4, 8, DEFER>(rsrc, ld, vlane, cb, q4, acc, qi, z, pend, pend_ptr);
    if (BORNVI_SYM_ABLATE < 2) {
      const int buf = (int)((t - t0) & 1);
#pragma unroll
      for (int c = 0; c < 4; ++c) zbuf[buf][wave][c * 64 + lane] = (sym_d2){z[c][0], z[c][1]};
      __syncthreads();
      const int tid = wave * 64 + lane;                 // columns cb + 2 tid, cb + 2 tid + 1
      if (SYM_WAVES == 4) {
        const sym_d2 sum = (zbuf[buf][0][tid] + zbuf[buf][1][tid]) + (zbuf[buf][2][tid] + zbuf[buf][3][tid]);
        *reinterpret_cast<double2*>(Z2s + cb + 2 * tid) = make_double2(sum.x, sum.y);
      } else {                                          // 512 threads, one column each
        const double* zb = reinterpret_cast<const double*>(&zbuf[buf][0][0]);
        double sum = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < SYM_WAVES; w2 += 2) sum += zb[w2 * 512 + tid] + zb[(w2 + 1) * 512 + tid];
        if (DEFER) { pend = sum; pend_ptr = Z2s + cb + tid; }
        else Z2s[cb + tid] = sum;
      }
    }
  }
  if (DEFER && t1 > t0) {
#if BORNVI_SYM_STORE_NT
    __builtin_nontemporal_store(pend, pend_ptr);
#else
    *pend_ptr = pend;
#endif
  }
  __syncthreads();      // the LDS buffers are reused by the next band of this workgroup
#pragma unroll
  for (int r = 0; r < SYM_ROWS; ++r) {
    const double v = wave_sum(acc[r]);
    if (lane == 0) yrow[i0 + r] = v;
  }
}

// workgroup = (band pair, column piece): band w and its mirror nbands-1-w -- a long and a short one, so every workgroup
// streams about the same number of bytes.  Strip-pair shard (several GPUs): this launch covers the pairs
// [pair_begin, pair_end); K_lo holds the rows of the bands [pair_begin, pair_end), K_hi those of the mirrored bands
// [nbands - pair_end, nbands - pair_begin).  (One GPU: all pairs, K_lo = K, K_hi = K + (nbands - npairs) SYM_BAND ld.)
__global__ __launch_bounds__(64 * SYM_WAVES, 2) void quadform_sym_kernel(const double* __restrict__ K_lo, const double* __restrict__ K_hi,
                                                                       long long ld, long long pair_begin, long long pair_end,
                                                                       const double* __restrict__ q, double* __restrict__ yrow,
                                                                       double* __restrict__ Z1, double* __restrict__ Z2,
                                                                       long long N, int nparts_log2) {
  __shared__ sym_d2 zbuf[2][SYM_WAVES][256];           // 2 x SYM_WAVES x 4 KiB: the waves' column partials of a trip
  const int lane = threadIdx.x & 63;
  // readfirstlane makes the wave index provably uniform, so q_i and all row offsets live in scalar registers
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long w = pair_begin + ((long long)blockIdx.x >> nparts_log2);
  const int nparts = 1 << nparts_log2;
  const int part = (int)(blockIdx.x & (unsigned)(nparts - 1));
  const long long nbands = N / SYM_BAND;
  if (w >= pair_end) return;
  double* __restrict__ yh = yrow + part * N;     // row partials of this piece (the reduce kernel adds them up)
  const long long S2 = nbands - 1 - w;
#pragma unroll 1
  for (int which = 0; which < 2; ++which) {      // the long band, then its short mirror (one copy of the code)
    if (which && S2 == w) break;
    const long long S = which ? S2 : w;
    const double* Kb = which ? K_hi + (S2 - (nbands - pair_end)) * SYM_BAND * ld : K_lo + (w - pair_begin) * SYM_BAND * ld;
    quadform_sym_band(Kb, ld, q, yh, Z1, Z2, N, S, lane, wave, part, nparts, zbuf);
  }
}

// y_j = sum over the pieces of yrow[j] + every column partial above row j: the bulk partials Z2 of the bands
// S < 4 (j / 512) and the near partials Z1 of the waves s in [16 (j / 512), j / 32).  64 columns per workgroup, the
// partials dealt to 4 waves and combined through LDS in wave order; also the per-workgroup partial of q . y.
// With a strip-pair shard only the bands [lo0, lo1) and [hi0, hi1) belong to this GPU: the sums run over those (and
// yrow counts only where column j's own band is one of them); y is then this GPU's PARTIAL of K q.
__global__ __launch_bounds__(256) void quadform_sym_reduce_kernel(const double* __restrict__ yrow, const double* __restrict__ Z1,
                                                                  const double* __restrict__ Z2, const double* __restrict__ q,
                                                                  double* __restrict__ y, double* __restrict__ partials,
                                                                  long long N, long long lo0, long long lo1, long long hi0,
                                                                  long long hi1, int nparts) {
  __shared__ double part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long j = (long long)blockIdx.x * 64 + lane;
  double acc = 0.0;
  if (j < N) {
    const long long smax = SYM_BP * (j / 512);      // bands whose bulk starts at or left of column j
#pragma unroll 1
    for (int rg = 0; rg < 2; ++rg) {
      const long long r0 = rg ? hi0 : lo0;
      const long long ns = rg ? (hi1 < smax ? hi1 : smax) : (lo1 < smax ? lo1 : smax);
      // eight loads in flight per lane (the adds keep their order: the sum is the same as the plain loop's)
      long long S = r0 + wave;
      for (; S + 28 < ns; S += 32) {
        double zv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) zv[u] = __builtin_nontemporal_load(Z2 + sym_z2_offset(S + 4 * u, N) + (j - sym_bulk_start(S + 4 * u, N)));
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += zv[u];
      }
      for (; S < ns; S += 4) acc += Z2[sym_z2_offset(S, N) + (j - sym_bulk_start(S, N))];
    }
    for (long long s = 16 * (j / 512) + wave; s < j / SYM_ROWS; s += 4) {     // waves whose NEAR range holds column j
      const long long S = s / SYM_WAVES;
      if ((S >= lo0 && S < lo1) || (S >= hi0 && S < hi1)) acc += Z1[s * SYM_NEAR + (j - (s + 1) * SYM_ROWS)];
    }
  }
  part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0) {
    double v = 0.0, contrib = 0.0;
    if (j < N) {
      const long long Sj = j / SYM_BAND;
      const bool own = (Sj >= lo0 && Sj < lo1) || (Sj >= hi0 && Sj < hi1);
      double yr = 0.0;
      if (own) for (int p = 0; p < nparts; ++p) yr += yrow[(long long)p * N + j];
      v = yr + ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]));
      if (y) y[j] = v;
      contrib = q[j] * v;
    }
    contrib = wave_sum(contrib);
    if (lane == 0) partials[blockIdx.x] = contrib;
  }
}

// workspace (doubles): yrow per column piece | per-workgroup partials of q . y | Z1 | Z2
static long long sym_ws_partials(long long N) { return (((N + 63) / 64) + 31) / 32 * 32; }
size_t quadform_sym_workspace_doubles(int n) {
  const long long N = 1ll << n;
  const size_t full = (size_t)quadform_partials(N) + 64;                 // full-matrix kernel
  if (n < SYM_MIN_N) return full;
  const long long nb = N / SYM_BAND;
  const long long z2 = sym_z2_offset(nb, N);
  const size_t band = (size_t)(SYM_MAX_PARTS * N) + (size_t)sym_ws_partials(N) + (size_t)(nb * SYM_WAVES * SYM_NEAR) + (size_t)(z2 > 0 ? z2 : 0) + 64;
  return band > full ? band : full;
}

// pairs [pair_begin, pair_end) of the N / SYM_BAND / 2 band pairs; K_lo / K_hi as in quadform_sym_kernel.
// ksd2 = sum_j q_j y_j over the (partial) y of this launch.
hipError_t launch_quadform_sym_pairs(int n, const double* K_lo, const double* K_hi, long long ld, long long pair_begin,
                                     long long pair_end, const double* q, double* y_or_null, double* ksd2, double* ws,
                                     hipStream_t st) {
  if (n < SYM_MIN_N) return hipErrorInvalidValue;
  const long long N = 1ll << n;
  const long long nb = N / SYM_BAND;
  double* yrow = ws;
  double* partials = ws + SYM_MAX_PARTS * N;
  const long long nred = (N + 63) / 64;
  double* Z1 = partials + sym_ws_partials(N);               // (16-byte aligned: every offset is even)
  double* Z2 = Z1 + nb * SYM_WAVES * SYM_NEAR;
  const long long npairs = pair_end - pair_begin;
  int parts_log2 = 0;                                       // enough column pieces for two workgroups per CU
  static const int min_wgs = [] { const char* e = getenv("BORNVI_SYM_MIN_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 512; }();
  while ((1 << parts_log2) < SYM_MAX_PARTS && (npairs << parts_log2) < min_wgs * 4 / SYM_WAVES) ++parts_log2;
  const int nparts = 1 << parts_log2;
  if (npairs > 0) {
    quadform_sym_kernel<<<(unsigned)(npairs << parts_log2), 64 * SYM_WAVES, 0, st>>>(K_lo, K_hi, ld, pair_begin, pair_end, q, yrow,
                                                                                  Z1, Z2, N, parts_log2);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  long long hi0 = nb - pair_end, hi1 = nb - pair_begin;
  if (hi0 < pair_end) hi0 = pair_end;                       // odd band count: the middle band is its own mirror
  quadform_sym_reduce_kernel<<<(unsigned)nred, 256, 0, st>>>(yrow, Z1, Z2, q, y_or_null, partials, N, pair_begin, pair_end, hi0, hi1, nparts);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  sum_partials_kernel<<<1, 256, 0, st>>>(partials, nred, ksd2);
  return hipGetLastError();
}

hipError_t launch_quadform_sym(int n, const double* K, long long ld, const double* q, double* y_or_null, double* ksd2,
                               double* ws, hipStream_t st) {
  const long long N = 1ll << n;
  if (n < SYM_MIN_N && ld != N) return hipErrorInvalidValue;
  if (n < SYM_MIN_N || (n < SYM_FULL_BELOW_N && ld == N))      // small: the full-matrix kernel (K is symmetric: same K q)
    return launch_quadform(n, K, 0, N, q, y_or_null, ksd2, ws, st);
  const long long nb = N / SYM_BAND;
  const long long npairs = (nb + 1) / 2;
  return launch_quadform_sym_pairs(n, K, K + (nb - npairs) * SYM_BAND * ld, ld, 0, npairs, q, y_or_null, ksd2, ws, st);
}

int quadform_sym_rows_per_strip() { return SYM_BAND; }
int quadform_sym_min_n() { return SYM_MIN_N; }

// ------------------------------------------------------------------------------------------------
// matrix-free y = K_p q.  The n+1 real vectors v_0 = q, v_{b+1} = s_b o q
// are packed two per complex128 state, pushed through K_base = M^{(x) n} by the circuit pass engine
// (real 2x2 butterflies act on re and im independently), then combined:
//   y = sum_b [ s_b o w_b - s_b o (u - flip_b u) - (w_b - flip_b w_b) + 2 (u - flip_b u) ],
//   u = K_base q, w_b = K_base (s_b o q).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kron_pack_kernel(int n, const double* __restrict__ S, const double* __restrict__ q,
                                                        double2* __restrict__ packed, double a, double* __restrict__ gate) {
  const long long N = 1ll << n;
  const int npk = (n + 2) / 2;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    gate[0] = 1.0; gate[1] = 0.0; gate[2] = a; gate[3] = 0.0;
    gate[4] = a; gate[5] = 0.0; gate[6] = 1.0; gate[7] = 0.0;
  }
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) {
    const double qi = q[i];
    for (int c = 0; c < npk; ++c) {
      const int v0 = 2 * c, v1 = 2 * c + 1;
      const double re = (v0 == 0) ? qi : S[i * n + (v0 - 1)] * qi;
      const double im = (v1 <= n) ? S[i * n + (v1 - 1)] * qi : 0.0;
      packed[c * N + i] = make_double2(re, im);
    }
  }
}

__global__ __launch_bounds__(256) void kron_combine_kernel(int n, const double* __restrict__ S, const double* __restrict__ q,
                                                           const double* __restrict__ packed, double* __restrict__ y,
                                                           double* __restrict__ partials) {
  __shared__ double red[256];
  const long long N = 1ll << n;
  const long long stride = (long long)gridDim.x * blockDim.x;
  double part = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) {
    const double ui = packed[i * 2];
    double acc = 0.0;
    for (int b = 0; b < n; ++b) {
      const long long f = i ^ (1ll << (n - 1 - b));
      const int v = b + 1;
      const double* wv = packed + ((long long)(v >> 1) * N) * 2 + (v & 1);
      const double wi = wv[i * 2], wf = wv[f * 2];
      const double du = ui - packed[f * 2];
      const double sb = S[i * n + b];
      acc += sb * wi - sb * du - (wi - wf) + 2.0 * du;
    }
    if (y) y[i] = acc;
    part += q[i] * acc;
  }
  red[threadIdx.x] = part;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}

size_t kron_partials(int n) {
  const long long N = 1ll << n;
  long long g = (N + 255) / 256;
  if (g > 2048) g = 2048;
  return (size_t)g;
}

hipError_t launch_kron_pack(int n, double length_scale, const double* S, const double* q, double* packed,
                            double* gate, hipStream_t st) {
  const double a = std::exp(-1.0 / ((double)n * length_scale));
  kron_pack_kernel<<<(unsigned)kron_partials(n), 256, 0, st>>>(n, S, q, (double2*)packed, a, gate);
  return hipGetLastError();
}

hipError_t launch_kron_combine(int n, const double* S, const double* q, const double* packed, double* y,
                               double* partials, double* ksd2, hipStream_t st) {
  const size_t g = kron_partials(n);
  kron_combine_kernel<<<(unsigned)g, 256, 0, st>>>(n, S, q, packed, y, partials);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  sum_partials_kernel<<<1, 256, 0, st>>>(partials, (long long)g, ksd2);
  return hipGetLastError();
}

hipError_t launch_score(const bornvi_bn_desc& bn, int n, double* S, double* pxz, hipStream_t st) {
  const long long N = 1ll << n;
  score_kernel<<<(unsigned)((N + 255) / 256), 256, 0, st>>>(bn, n, S, pxz);
  return hipGetLastError();
}

}  // namespace bornvi
