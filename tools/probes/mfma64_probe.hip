// How long does v_mfma_f64_16x16x4_f64 occupy a SIMD on gfx950, and what can issue beside it?  One wave per SIMD
// (256-thread workgroups, one per CU), ITER trips of a loop with NM independent MFMAs and NV other instructions of one
// kind interleaved; prints cycles per trip (s_memtime is at 100 MHz: the kernel's duration from HIP events and the
// clock come out of the ratio).   hipcc -O3 --offload-arch=gfx950 mfma64_probe.hip -o mfma64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NM, int KIND, int NV>
__global__ __launch_bounds__(256) void probe(double* out, int iters, double seed) {
  d4 acc[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) acc[i] = d4{seed, seed, seed, seed};
  double a = seed + threadIdx.x, b = seed * 0.5;
  double v[8];
  int u[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { v[i] = seed * (i + 1); u[i] = (int)threadIdx.x + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      acc[m % 6] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[m % 6], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int s = (m * NV + k) & 7;
        if (KIND == 1) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(v[s]) : "v"(b));
        if (KIND == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[s]) : "v"(b));
        if (KIND == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[s]) : "v"(u[(s + 1) & 7]));
        if (KIND == 4) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(u[s]) : "v"(u[(s + 1) & 7]));
      }
    }
  }
  double r = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 8; ++i) r += v[i] + u[i];
  if (r == 1.2345e-300) out[0] = r;
}

template <int NM, int KIND, int NV>
static void run(const char* name, double* d) {
  const int iters = 200000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<NM, KIND, NV><<<256, 256>>>(d, 100, 1.0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<NM, KIND, NV><<<256, 256>>>(d, iters, 1.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double ns_per_trip = ms * 1e6 / iters;
  printf("%-44s NM=%d NV=%d : %8.1f ns per trip = %7.1f cycles at 2.4 GHz", name, NM, NV, ns_per_trip, ns_per_trip * 2.4);
  if (NM) printf("  (%.1f cycles per MFMA", ns_per_trip * 2.4 / NM), printf(", %.1f TFLOP/s fp64 on 1024 SIMDs)", NM * 2048.0 * 1024 / ns_per_trip * 1e-3);
  printf("\n");
}

// two waves per SIMD (512-thread workgroups): waves 0..3 issue MFMAs only, waves 4..7 vector instructions only
// (MODE bit 0: the MFMA waves run, bit 1: the vector waves run).  Do the two overlap ACROSS waves?
template <int MODE, int KIND>
__global__ __launch_bounds__(512) void probe2(double* out, int iters, double seed) {
  const int wave = threadIdx.x >> 6;
  double r = 0.0;
  if (wave < 4) {
    if (!(MODE & 1)) return;
    d4 acc[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[i] = d4{seed, seed, seed, seed};
    double a = seed + threadIdx.x, b = seed * 0.5;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 6; ++m) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[m], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    if (!(MODE & 2)) return;
    double v[8];
    int u[8];
    const double b = seed * 0.5;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = seed * (i + 1); u[i] = (int)threadIdx.x + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 96; ++k) {      // 96 instructions ~ the time of 6 MFMAs
        const int s = k & 7;
        if (KIND == 1) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(v[s]) : "v"(b));
        if (KIND == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[s]) : "v"(u[(s + 1) & 7]));
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) r += v[i] + u[i];
  }
  if (r == 1.2345e-300) out[0] = r;
}

template <int MODE, int KIND>
static void run2(const char* name, double* d) {
  const int iters = 200000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  probe2<MODE, KIND><<<256, 512>>>(d, 1000, 1.0);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  probe2<MODE, KIND><<<256, 512>>>(d, iters, 1.0);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-60s : %8.1f ns per trip (%.1f ms in all)\n", name, ms * 1e6 / iters, ms);
}

int main() {
  double* d;
  hipMalloc(&d, 4096);
  run<6, 0, 0>("MFMA f64 16x16x4 back to back", d);
  run<0 + 6, 1, 4>("MFMA + 4 v_fma_f64 each", d);
  run<6, 1, 8>("MFMA + 8 v_fma_f64 each", d);
  run<6, 1, 16>("MFMA + 16 v_fma_f64 each", d);
  run<6, 2, 8>("MFMA + 8 v_add_f64 each", d);
  run<6, 3, 8>("MFMA + 8 v_add_u32 each", d);
  run<6, 3, 16>("MFMA + 16 v_add_u32 each", d);
  run<6, 4, 16>("MFMA + 16 v_fma_f32 each", d);
  run2<1, 1>("two waves per SIMD: 6 MFMAs per trip, other wave idle", d);
  run2<2, 1>("two waves per SIMD: 96 v_fma_f64 per trip, other wave idle", d);
  run2<3, 1>("two waves per SIMD: 6 MFMAs | 96 v_fma_f64 on the other wave", d);
  run2<2, 3>("two waves per SIMD: 96 v_add_u32 per trip, other wave idle", d);
  run2<3, 3>("two waves per SIMD: 6 MFMAs | 96 v_add_u32 on the other wave", d);
  return 0;
}
