// Batched dense contraction Y = K_p Q^T on the matrix cores (gfx950): the GEMM-shaped piece of the KSD path.
//
// Replaces, for B > 1 probability vectors at once, the accumulation  sum_ij q_i q_j k_p(z_i, z_j)  of
// ksd_vi_quantum.py:123-145 evaluated B times: Y[b, i] = sum_j K[i, j] Q[b, j], ksd2[b] = sum_i Q[b, i] Y[b, i].
// (With Q = the 2P + 1 rows of a parameter-shift batch this is the KSD at every shifted point, the diagnostic
// SURVEY.md section 0.5 describes; the training step itself needs B = 1 and stays on the HBM-bound kernels.)
//
// One pass over K: 8 * 4^n bytes from HBM (34.4 GB at n = 16), 2 * 4^n * B flop (4.96 TFLOP at B = 577) ->
// MFMA-bound from B of about 40 on (fp64 matrix peak 78.6 TFLOP/s).  C = A B^T with A = K [N x N] and B = Q [B x N],
// both with the reduction index contiguous; v_mfma_f64_16x16x4_f64 on 64 x 64 wave tiles:
//   workgroup = 8 waves (4 along i x 2 along b) = 256 rows of K x 128 vectors; k-step 16; LDS double-buffered
//   (one barrier per k-step), next k-step's global loads in flight in registers during the MFMAs;
//   LDS rows padded to 18 doubles: the 16 rows x 2 k a half-wave reads with ds_read_b64 hit 32 distinct 8-byte slots;
//   workgroups that share a block of K rows (the B / 128 vector blocks) are adjacent in the grid: K comes from HBM
//   once and from L2 after that.
// Fragment layout of the f64 MFMA (cdna_hip_programming.md section 3): A[row = lane & 15][k = lane >> 4],
// B[k = lane >> 4][col = lane & 15], D[row = (lane >> 4) + 4 r][col = lane & 15], r = 0..3.
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace bornvi {

namespace {
constexpr int QB_BM = 256, QB_BN = 128, QB_BK = 16, QB_PITCH = 18, QB_THREADS = 512;
typedef double qb_d4 __attribute__((ext_vector_type(4)));
typedef double qb_d2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(QB_THREADS) void quadform_batched_kernel(const double* __restrict__ K, const double* __restrict__ Q,
                                                                     double* __restrict__ Y, long long N, long long ld, int B) {
  extern __shared__ double qb_lds[];
  double* __restrict__ As = qb_lds;                                   // [2][QB_BM][QB_PITCH]
  double* __restrict__ Bs = qb_lds + 2 * QB_BM * QB_PITCH;            // [2][QB_BN][QB_PITCH]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wi = wave & 3, wb = wave >> 2;                            // wave tile: rows wi * 64, vectors wb * 64
  const long long i_blk = (long long)blockIdx.y * QB_BM;
  const int b_blk = blockIdx.x * QB_BN;
  // global -> register staging: 16 bytes per thread and load; A: 4 loads (rows t/8 + 64 u), B: 2 loads (vectors t/8 + 64 u)
  const int lrow = t >> 3, lk = (t & 7) * 2;
  const double* __restrict__ Ag = K + (i_blk + lrow) * ld + lk;        // (ld: row pitch of K in doubles, >= N)
  const double* __restrict__ Bg = Q + (long long)(b_blk + lrow) * N + lk;
  bool bok[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) bok[u] = b_blk + lrow + 64 * u < B;    // vectors past B read as zero
  qb_d2 ra[4], rb[2];
  auto load_tiles = [&](long long k0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const double* p = Ag + (long long)(64 * u) * ld + k0;
      ra[u].x = __builtin_nontemporal_load(p); ra[u].y = __builtin_nontemporal_load(p + 1);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (bok[u]) rb[u] = *reinterpret_cast<const qb_d2*>(Bg + (long long)(64 * u) * N + k0);
      else rb[u] = (qb_d2){0.0, 0.0};
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 4; ++u) *reinterpret_cast<qb_d2*>(As + ((buf * QB_BM + lrow + 64 * u) * QB_PITCH + lk)) = ra[u];
#pragma unroll
    for (int u = 0; u < 2; ++u) *reinterpret_cast<qb_d2*>(Bs + ((buf * QB_BN + lrow + 64 * u) * QB_PITCH + lk)) = rb[u];
  };
  qb_d4 acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = (qb_d4){0.0, 0.0, 0.0, 0.0};
  const int fr = lane & 15, fk = lane >> 4;
  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  const long long nk = N / QB_BK;
#pragma unroll 1
  for (long long kt = 0; kt < nk; ++kt) {
    const int cur = (int)(kt & 1);
    if (kt + 1 < nk) load_tiles((kt + 1) * QB_BK);                    // in flight during this k-step's MFMAs
    const double* __restrict__ Ac = As + (cur * QB_BM + wi * 64 + fr) * QB_PITCH + fk;
    const double* __restrict__ Bc = Bs + (cur * QB_BN + wb * 64 + fr) * QB_PITCH + fk;
#pragma unroll
    for (int ks = 0; ks < QB_BK / 4; ++ks) {
      double a[4], b[4];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) a[mi] = Ac[mi * 16 * QB_PITCH + ks * 4];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) b[ni] = Bc[ni * 16 * QB_PITCH + ks * 4];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tiles(cur ^ 1);
    __syncthreads();
  }
  // D[row = fk + 4 r][col = fr] of tile (mi, ni): Y[b][i] with b = b_blk + wb * 64 + ni * 16 + fr, i = i_blk + wi * 64 + mi * 16 + fk + 4 r
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int b = b_blk + wb * 64 + ni * 16 + fr;
    if (b >= B) continue;
    double* __restrict__ Yb = Y + (long long)b * N + i_blk + wi * 64 + fk;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) Yb[mi * 16 + 4 * r] = acc[mi][ni][r];
  }
}

// ksd2[b] = Q[b] . Y[b]; one workgroup per vector, fixed-order tree: deterministic
__global__ __launch_bounds__(1024) void rowdot_kernel(const double* __restrict__ Q, const double* __restrict__ Y, long long N,
                                                      double* __restrict__ out) {
  __shared__ double red[1024];
  const double* q = Q + (long long)blockIdx.x * N;
  const double* y = Y + (long long)blockIdx.x * N;
  double acc = 0.0;
  for (long long i = threadIdx.x * 2; i < N; i += 2048) {
    const qb_d2 a = *reinterpret_cast<const qb_d2*>(q + i), b = *reinterpret_cast<const qb_d2*>(y + i);
    acc += a.x * b.x + a.y * b.y;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}
}  // namespace

bool quadform_batched_supported(int n, int B) { return n >= 8 && n <= 17 && B >= 2; }

hipError_t launch_quadform_batched(int n, const double* K, long long ld, const double* Q, int B, double* Y, double* ksd2, hipStream_t st) {
  const long long N = 1ll << n;
  const size_t lds = (size_t)2 * (QB_BM + QB_BN) * QB_PITCH * sizeof(double);
  // (the attribute is per device and cheap to set: no process-wide "done" flag that a second device or a second thread
  // could find set before its own call went through)
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(quadform_batched_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  // x = vector block (fastest): the workgroups that share a block of K rows are dispatched together
  dim3 grid((unsigned)((B + QB_BN - 1) / QB_BN), (unsigned)(N / QB_BM));
  quadform_batched_kernel<<<grid, QB_THREADS, lds, st>>>(K, Q, Y, N, ld, B);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  rowdot_kernel<<<(unsigned)B, 1024, 0, st>>>(Q, Y, N, ksd2);
  return hipGetLastError();
}

}  // namespace bornvi
