// What does the Gram builder's store pattern cost by itself?  A 2^16 x 2^16 matrix of doubles (pitch 2^16 + 32) written
// with (A) the shipped pattern: a wave stores 16-column tiles, 4 rows x 128 bytes per instruction, the four waves of a
// workgroup on adjacent tiles; (B) a wave stores four adjacent tiles of a row tile back to back (512 contiguous bytes per
// row); (C) like A with plain instead of non-temporal stores; (D) a plain fill (16 bytes per lane, 1 KiB per instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr long long N = 1ll << 16, LD = N + 32, COLS = 2048, ROWS = 64;

template <int MODE>
__global__ __launch_bounds__(256) void pattern(double* __restrict__ K, double v) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ar = lane & 15, ak = lane >> 4;
  const long long i_blk = (long long)blockIdx.y * ROWS, j_chunk = (long long)blockIdx.x * COLS;
  double* Kb = K + i_blk * LD;
  if (MODE == 1) {
    for (long long j0 = j_chunk + wave * 64; j0 < j_chunk + COLS; j0 += 256)
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            __builtin_nontemporal_store(v + r, Kb + (rt * 16 + ak + 4 * r) * LD + j0 + 16 * c4 + ar);
  } else {
    for (long long j0 = j_chunk + wave * 16; j0 < j_chunk + COLS; j0 += 64)
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double* p = Kb + (rt * 16 + ak + 4 * r) * LD + j0 + ar;
          if (MODE == 2) *p = v + r; else __builtin_nontemporal_store(v + r, p);
        }
  }
}

__global__ __launch_bounds__(256) void fill(double2* __restrict__ K, long long n2, double v) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long long)gridDim.x * 256) K[i] = make_double2(v, v);
}

template <typename F> static float timed(F f) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  f(); (void)hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    (void)hipEventRecord(a); f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
  }
  return best;
}

int main() {
  double* K; if (hipMalloc(&K, N * LD * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  dim3 grid((unsigned)(N / COLS), (unsigned)(N / ROWS));
  const double gb = N * N * 8 / 1e9;
  float t;
  t = timed([&] { pattern<0><<<grid, 256>>>(K, 1.0); }); printf("A shipped pattern, non-temporal : %.2f ms = %.0f GB/s\n", t, gb / t * 1e3);
  t = timed([&] { pattern<1><<<grid, 256>>>(K, 1.0); }); printf("B 512 bytes of a row per wave   : %.2f ms = %.0f GB/s\n", t, gb / t * 1e3);
  t = timed([&] { pattern<2><<<grid, 256>>>(K, 1.0); }); printf("C shipped pattern, plain stores : %.2f ms = %.0f GB/s\n", t, gb / t * 1e3);
  t = timed([&] { fill<<<256 * 16, 256>>>((double2*)K, N * LD / 2, 1.0); }); printf("D plain fill                    : %.2f ms = %.0f GB/s\n", t, N * LD * 8 / 1e9 / t * 1e3);
  return 0;
}
