// Probe (not product code): what read+write rate does the MI355X sustain on a ping-pong working set that fits the
// 256 MiB Infinity Cache, against one that does not?  Decides whether a depth-first pass schedule of the circuit
// engine (states re-read while still cache-resident) can pay.  Each 128-thread workgroup owns T tile pairs of
// 32 KiB: per pass it reads tile t of buffer A (16 x 16 B per lane) and writes tile t of buffer B, then swaps.
// 1024 workgroups x T x 64 KiB = T x 64 MiB working set (8 T MiB per XCD, above its 4 MiB L2 from T = 1 on).
// flavour 0: plain loads/stores, 1: nt, 2: sc1 (the cross-workgroup-visible form a depth-first schedule would need).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2_t __attribute__((ext_vector_type(2)));

template <int FL>
__device__ __forceinline__ void ld16(d2_t& dst, const char* p) {
  if (FL == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
  if (FL == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(p) : "memory");
  if (FL == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(dst) : "v"(p) : "memory");
}
template <int FL>
__device__ __forceinline__ void st16(char* p, d2_t v) {
  if (FL == 0) asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(v) : "memory");
  if (FL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p), "v"(v) : "memory");
  if (FL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}

template <int FL>
__global__ __launch_bounds__(128) void pingpong(char* A, char* B, int T, int passes) {
  const size_t tile_bytes = 32768;
  char* a = A + (size_t)blockIdx.x * T * tile_bytes;
  char* b = B + (size_t)blockIdx.x * T * tile_bytes;
  for (int p = 0; p < passes; ++p) {
    for (int t = 0; t < T; ++t) {
      d2_t v[16];
      const char* src = a + (size_t)t * tile_bytes + threadIdx.x * 16;
      char* dst = b + (size_t)t * tile_bytes + ((threadIdx.x * 16) ^ 0x400);   // a different 1 KiB run on the way out
#pragma unroll
      for (int i = 0; i < 16; ++i) ld16<FL>(v[i], src + i * 2048);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 16; ++i) { v[i].x += 1.0; st16<FL>(dst + i * 2048, v[i]); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    char* tmp = a; a = b; b = tmp;
  }
}

int main() {
  const int wgs = 1024;
  const int Ts[] = {1, 2, 3, 4, 6, 16, 32};
  size_t maxbytes = (size_t)wgs * 32 * 32768;
  char *A, *B;
  if (hipMalloc(&A, maxbytes) != hipSuccess || hipMalloc(&B, maxbytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(A, 0, maxbytes); hipMemset(B, 0, maxbytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int fl = 0; fl < 3; ++fl)
    for (int T : Ts) {
      const int passes = 4096 / T > 64 ? 64 * 4 / (T > 4 ? 4 : T) : 16;   // a few ms per launch
      auto launch = [&](int P) {
        if (fl == 0) pingpong<0><<<wgs, 128>>>(A, B, T, P);
        if (fl == 1) pingpong<1><<<wgs, 128>>>(A, B, T, P);
        if (fl == 2) pingpong<2><<<wgs, 128>>>(A, B, T, P);
      };
      launch(4);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      launch(passes);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      const double bytes = 2.0 * wgs * T * 32768.0 * passes;
      printf("flavour %d  T %2d  working set %5d MiB  passes %3d  %.3f ms  %.0f GB/s (read+write)\n", fl, T, T * 64, passes, ms,
             bytes / (ms * 1e-3) / 1e9);
      fflush(stdout);
    }
  return 0;
}
