// Probe (not product code): read rate of the upper triangle of a 2^16 x 2^16 fp64 matrix (17.2 GB) in three access
// shapes, on several 32 GiB allocations held at once:
//   A "strip"      the round-1/2 symmetric contraction's shape: a wave owns 32 rows (512 KiB apart) and sweeps their
//                  columns 4 KiB per row and trip, 8 rows x 4 chunks in flight;
//   B "packed"     the same bytes if each strip's data were stored contiguously (strip-major blocks of 32 rows x 512
//                  columns = 128 KiB): a wave streams its strip sequentially, same 32 loads in flight;
//   C "linear"     plain grid-stride stream over the same number of bytes.
// Answers: is the slow / fast allocation effect a property of the strided shape (TLB reach, DRAM pages) or of the memory?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2_t __attribute__((ext_vector_type(2)));
constexpr long long N = 65536, ROWS = 32;

__device__ __forceinline__ d2_t ldnt(const double* p) {
  d2_t v;
  v.x = __builtin_nontemporal_load(p);
  v.y = __builtin_nontemporal_load(p + 1);
  return v;
}

__global__ void fill_random(double* K, long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    unsigned long long h = (unsigned long long)i * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    K[i] = (double)(long long)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5;
  }
}

// the contraction's arithmetic on the same loads: 32 row accumulators and 8 column accumulators, 4 fp64 FMAs per 16 bytes
__global__ __launch_bounds__(256) void shape_strip_fma(const double* __restrict__ K, long long ld, const double* __restrict__ q,
                                                       double* __restrict__ out, int store_z) {
  const int lane = threadIdx.x & 63;
  const long long wi = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long pair = wi >> 1, part = wi & 1, ns = N / ROWS;
  double tot = 0.0;
  for (int which = 0; which < 2; ++which) {
    const long long s = which ? ns - 1 - pair : pair;
    const long long i0 = s * ROWS;
    const long long cstart = ((i0 + 511) / 512) * 512;
    const long long half = ((N - cstart) / 1024) * 512;
    const long long c0 = cstart + part * half, c1 = part ? N : c0 + half;
    const double* Kr = K + i0 * ld;
    double acc[ROWS], qi[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) { acc[r] = 0.0; qi[r] = q[i0 + r]; }
    for (long long cb = c0; cb + 512 <= c1; cb += 512) {
      d2_t q4[4];
      double z[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
#pragma unroll
      for (int c = 0; c < 4; ++c) q4[c] = *reinterpret_cast<const d2_t*>(q + cb + c * 128 + lane * 2);
#pragma unroll
      for (int r0 = 0; r0 < ROWS; r0 += 8) {
        d2_t v[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int c = 0; c < 4; ++c) v[u][c] = ldnt(Kr + (r0 + u) * ld + cb + c * 128 + lane * 2);
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            acc[r0 + u] = fma(v[u][c].x, q4[c].x, fma(v[u][c].y, q4[c].y, acc[r0 + u]));
            z[c][0] = fma(v[u][c].x, qi[r0 + u], z[c][0]);
            z[c][1] = fma(v[u][c].y, qi[r0 + u], z[c][1]);
          }
      }
      if (store_z) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          *reinterpret_cast<d2_t*>(out + (1 << 16) + ((wi * 2 + which) % 4096) * 65536 + (cb % 65536) + c * 128 + lane * 2) = (d2_t){z[c][0], z[c][1]};
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) tot += z[c][0] + z[c][1];
      }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) tot += acc[r];
  }
  if (tot == 1.2345e300) out[wi] = tot;
}

// strips s and ns-1-s, part of 2: columns [c0, c1) in trips of 512
__global__ __launch_bounds__(256) void shape_strip(const double* __restrict__ K, long long ld, double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long long wi = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long pair = wi >> 1, part = wi & 1, ns = N / ROWS;
  double acc = 0.0;
  for (int which = 0; which < 2; ++which) {
    const long long s = which ? ns - 1 - pair : pair;
    const long long i0 = s * ROWS;
    const long long cstart = ((i0 + 511) / 512) * 512;
    const long long half = ((N - cstart) / 1024) * 512;
    const long long c0 = cstart + part * half, c1 = part ? N : c0 + half;
    const double* Kr = K + i0 * ld;
    for (long long cb = c0; cb + 512 <= c1; cb += 512)
      for (int r0 = 0; r0 < ROWS; r0 += 8) {
        d2_t v[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int c = 0; c < 4; ++c) v[u][c] = ldnt(Kr + (r0 + u) * ld + cb + c * 128 + lane * 2);
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc += v[u][c].x + v[u][c].y;
      }
  }
  if (acc == 1.2345e300) out[wi] = acc;
}

// the same bytes, strip-major: wave wi streams `blocks` consecutive 128 KiB blocks starting at block `first`
__global__ __launch_bounds__(256) void shape_packed(const double* __restrict__ K, const long long* __restrict__ first,
                                                    const long long* __restrict__ count, double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long long wi = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  double acc = 0.0;
  for (int which = 0; which < 2; ++which) {
    const double* B = K + first[2 * wi + which] * 16384;      // 128 KiB = 16384 doubles per block
    const long long nb = count[2 * wi + which];
    for (long long b = 0; b < nb; ++b)
      for (int r0 = 0; r0 < ROWS; r0 += 8) {
        d2_t v[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int c = 0; c < 4; ++c) v[u][c] = ldnt(B + b * 16384 + (r0 + u) * 512 + c * 128 + lane * 2);
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc += v[u][c].x + v[u][c].y;
      }
  }
  if (acc == 1.2345e300) out[wi] = acc;
}

__global__ __launch_bounds__(256) void shape_linear(const double* __restrict__ K, long long n2, double* __restrict__ out) {
  const long long stride = (long long)gridDim.x * 256;
  double acc = 0.0;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n2; i += 4 * stride) {
    d2_t v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = ldnt(K + 2 * (i + u * stride));
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += v[u].x + v[u].y;
  }
  if (acc == 1.2345e300) out[i & 1023] = acc;
}

int main(int argc, char** argv) {
  const int nalloc = argc > 1 ? atoi(argv[1]) : 3;
  const long long pad = argc > 2 ? atoll(argv[2]) : 0;
  const int fill = argc > 3 ? atoi(argv[3]) : 0;
  const long long ld = N + pad;
  const long long ns = N / ROWS, npairs = ns / 2, nwaves = npairs * 2;
  // packed layout: strip s holds blocks for column blocks cb >= ceil(i0 / 512) ... 127; strips in order
  std::vector<long long> first(2 * nwaves), count(2 * nwaves), strip_first(ns), strip_blocks(ns);
  long long tot = 0;
  for (long long s = 0; s < ns; ++s) { strip_first[s] = tot; strip_blocks[s] = 128 - (s * ROWS + 511) / 512; tot += strip_blocks[s]; }
  double bytes = 0;
  for (long long wi = 0; wi < nwaves; ++wi) {
    const long long pair = wi >> 1, part = wi & 1;
    for (int which = 0; which < 2; ++which) {
      const long long s = which ? ns - 1 - pair : pair;
      const long long half = strip_blocks[s] / 2;
      first[2 * wi + which] = strip_first[s] + part * half;
      count[2 * wi + which] = part ? strip_blocks[s] - half : half;
      bytes += (double)count[2 * wi + which] * 131072.0;
    }
  }
  printf("triangle blocks %lld = %.2f GB; pitch %lld doubles\n", tot, bytes / 1e9, ld);
  long long *dfirst, *dcount;
  double* dout;
  (void)hipMalloc(&dfirst, first.size() * 8); (void)hipMalloc(&dcount, count.size() * 8); (void)hipMalloc(&dout, ((size_t)1 << 19) + (size_t)4096 * 65536 * 8);
  double* dq;
  (void)hipMalloc(&dq, N * 8);
  fill_random<<<256, 256>>>(dq, N);
  (void)hipMemcpy(dfirst, first.data(), first.size() * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(dcount, count.data(), count.size() * 8, hipMemcpyHostToDevice);
  std::vector<double*> Ks;
  for (int a = 0; a < nalloc; ++a) {
    double* K;
    if (hipMalloc(&K, (size_t)N * ld * 8) != hipSuccess) { printf("alloc %d failed\n", a); break; }
    if (fill) fill_random<<<4096, 256>>>(K, N * ld); else (void)hipMemset(K, 0, (size_t)N * ld * 8);
    Ks.push_back(K);
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto timeit = [&](auto&& launch) {
    launch(); (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    return best;
  };
  for (size_t a = 0; a < Ks.size(); ++a) {
    double* K = Ks[a];
    const float ta = timeit([&] { shape_strip<<<(unsigned)(nwaves / 4), 256>>>(K, ld, dout); });
    const float tb = timeit([&] { shape_packed<<<(unsigned)(nwaves / 4), 256>>>(K, dfirst, dcount, dout); });
    const float tc = timeit([&] { shape_linear<<<4096, 256>>>(K, (long long)(bytes / 16), dout); });
    const float td = timeit([&] { shape_strip_fma<<<(unsigned)(nwaves / 4), 256>>>(K, ld, dq, dout, 0); });
    const float te = timeit([&] { shape_strip_fma<<<(unsigned)(nwaves / 4), 256>>>(K, ld, dq, dout, 1); });
    printf("alloc %zu at %p fill %d: strip %.3f ms (%.0f GB/s)  packed %.3f (%.0f)  linear %.3f (%.0f)  strip+fma %.3f (%.0f)  strip+fma+z %.3f (%.0f)\n",
           a, (void*)K, fill, ta, bytes / ta / 1e6, tb, bytes / tb / 1e6, tc, bytes / tc / 1e6, td, bytes / td / 1e6, te, bytes / te / 1e6);
    fflush(stdout);
  }
  return 0;
}
