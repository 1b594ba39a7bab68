// The HBM side of a circuit pass by itself, in several shapes, on a batch of 385 states of 2^20 complex128 (6.5 GB in,
// 6.5 GB out): what separates the pass kernel's 5.1 TB/s (no stages) from the 5.7-5.9 TB/s of an out-of-place
// elementwise kernel?
//   A  many small workgroups (256 threads, 4 x 16 B per thread, loads then stores), plain accesses  [torch's shape]
//   B  the same with non-temporal loads and stores
//   C  persistent, one 1024-thread workgroup per CU, 128 KiB contiguous per trip, next trip's loads issued before this
//      trip's stores, non-temporal
//   D  like C, but a trip's 128 KiB = 128 runs of 1 KiB at a stride of 128 KiB (the pass kernel's tile), non-temporal
//   E  like D with plain accesses
//   F  like D, 2 workgroups of 512 threads per CU (64 KiB tiles)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2_t __attribute__((ext_vector_type(2)));
constexpr long long STATE = 1ll << 20, BATCH = 385, TOTAL = STATE * BATCH;   // complex128 elements (16 B each)

template <bool NT> __device__ __forceinline__ d2_t ld(const d2_t* p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}
template <bool NT> __device__ __forceinline__ void st(d2_t* p, d2_t v) {
  if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

template <bool NT>
__global__ __launch_bounds__(256) void small_blocks(const d2_t* __restrict__ in, d2_t* __restrict__ out) {
  const long long base = (long long)blockIdx.x * 1024 + threadIdx.x;
  d2_t v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) v[u] = ld<NT>(in + base + u * 256);
#pragma unroll
  for (int u = 0; u < 4; ++u) { v[u].x += 1.0; st<NT>(out + base + u * 256, v[u]); }
}

// T threads per workgroup, tile = 8 T elements.  Shape of a side (read / write): RUN = log2 of the contiguous run in
// elements (16 B each); RUN = 0: the whole tile contiguous; else element e of the tile sits at
// (e >> RUN) * (TILES_PER_STATE << RUN) + g << RUN + (e & (2^RUN - 1)) of its state (g = tile index inside the state).
template <int T, int RRUN, int WRUN, bool NT>
__global__ __launch_bounds__(T) void persistent2(const d2_t* __restrict__ in, d2_t* __restrict__ out) {
  constexpr long long TILE = 8ll * T;
  constexpr long long TILES_PER_STATE = STATE / TILE;
  const long long ntiles = TOTAL / TILE;
  auto addr_run = [&](long long tile, int u, int run) -> long long {
    const long long e = (long long)u * T + threadIdx.x;          // element of the tile
    if (run == 0) return tile * TILE + e;
    const long long b = tile / TILES_PER_STATE, g = tile % TILES_PER_STATE;
    return b * STATE + (e >> run) * (TILES_PER_STATE << run) + (g << run) + (e & ((1ll << run) - 1));
  };
  d2_t v[8], w[8];
  long long t = blockIdx.x;
  if (t >= ntiles) return;
#pragma unroll
  for (int u = 0; u < 8; ++u) v[u] = ld<NT>(in + addr_run(t, u, RRUN));
  for (; t < ntiles; t += gridDim.x) {
    const long long tn = t + gridDim.x;
    if (tn < ntiles) {
#pragma unroll
      for (int u = 0; u < 8; ++u) w[u] = ld<NT>(in + addr_run(tn, u, RRUN));
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { v[u].x += 1.0; st<NT>(out + addr_run(t, u, WRUN), v[u]); }
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = w[u];
  }
}

template <int T, bool STRIDED, bool NT>
__global__ __launch_bounds__(T) void persistent(const d2_t* __restrict__ in, d2_t* __restrict__ out) {
  constexpr long long TILE = 8ll * T;
  constexpr long long TILES_PER_STATE = STATE / TILE;
  const long long ntiles = TOTAL / TILE;
  auto addr = [&](long long tile, int u) -> long long {
    const long long e = (long long)u * T + threadIdx.x;          // element of the tile
    if (!STRIDED) return tile * TILE + e;
    const long long b = tile / TILES_PER_STATE, g = tile % TILES_PER_STATE;
    return b * STATE + (e >> 6) * (TILES_PER_STATE * 64) + g * 64 + (e & 63);
  };
  d2_t v[8], w[8];
  long long t = blockIdx.x;
  if (t >= ntiles) return;
#pragma unroll
  for (int u = 0; u < 8; ++u) v[u] = ld<NT>(in + addr(t, u));
  for (; t < ntiles; t += gridDim.x) {
    const long long tn = t + gridDim.x;
    if (tn < ntiles) {
#pragma unroll
      for (int u = 0; u < 8; ++u) w[u] = ld<NT>(in + addr(tn, u));
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { v[u].x += 1.0; st<NT>(out + addr(t, u), v[u]); }
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = w[u];
  }
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
  d2_t *in, *out;
  if (hipMalloc(&in, TOTAL * 16) != hipSuccess || hipMalloc(&out, TOTAL * 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMemset(in, 0, TOTAL * 16);
  const double gb = 2.0 * TOTAL * 16 / 1e9;
  float t;
  t = timed([&] { small_blocks<false><<<(unsigned)(TOTAL / 1024), 256>>>(in, out); }); printf("A small blocks, plain          : %.2f ms = %.0f GB/s\n", t, gb / t * 1e3);
  t = timed([&] { small_blocks<true><<<(unsigned)(TOTAL / 1024), 256>>>(in, out); });  printf("B small blocks, non-temporal   : %.2f ms = %.0f GB/s\n", t, gb / t * 1e3);
  t = timed([&] { persistent<1024, false, true><<<256, 1024>>>(in, out); });            printf("C persistent contiguous, nt    : %.2f ms = %.0f GB/s\n", t, gb / t * 1e3);
  t = timed([&] { persistent<1024, true, true><<<256, 1024>>>(in, out); });             printf("D persistent 1 KiB runs, nt    : %.2f ms = %.0f GB/s\n", t, gb / t * 1e3);
  t = timed([&] { persistent<1024, true, false><<<256, 1024>>>(in, out); });            printf("E persistent 1 KiB runs, plain : %.2f ms = %.0f GB/s\n", t, gb / t * 1e3);
  t = timed([&] { persistent<512, true, true><<<512, 512>>>(in, out); });               printf("F 2 x 512 threads per CU, nt   : %.2f ms = %.0f GB/s\n", t, gb / t * 1e3);
  t = timed([&] { persistent<256, true, true><<<2048, 256>>>(in, out); });              printf("G 8 x 256 threads per CU, nt   : %.2f ms = %.0f GB/s\n", t, gb / t * 1e3);
  t = timed([&] { persistent<256, false, false><<<2048, 256>>>(in, out); });            printf("H 8 x 256 contiguous, plain    : %.2f ms = %.0f GB/s\n", t, gb / t * 1e3);
#define RW(R_, W_) t = timed([&] { persistent2<1024, R_, W_, true><<<256, 1024>>>(in, out); }); \
  printf("persistent, read runs 2^%d, write runs 2^%d elements (0 = whole tile), nt: %.2f ms = %.0f GB/s\n", R_, W_, t, gb / t * 1e3);
  RW(0, 6) RW(6, 0) RW(0, 0) RW(6, 6) RW(8, 8) RW(10, 10) RW(11, 11) RW(12, 12) RW(0, 8) RW(0, 10) RW(10, 6)
  return 0;
}
