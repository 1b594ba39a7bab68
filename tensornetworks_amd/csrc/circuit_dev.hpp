// Device helpers shared by the two persistent circuit kernels (kernels_circuit.hip: 16 amplitudes per thread;
// kernels_circuit8.hip: 8 amplitudes per thread): the in-place 2x2 complex gate and the vector memory ops the compiler
// does not track.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace bornvi {

typedef double d2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void gate_pair_inplace(double& x0r, double& x0i, double& x1r, double& x1i,
                                                  const double (&U)[8]) {
  double t0, t1, t2, t3;
  asm("v_mul_f64 %4, %9, %1\n\t"          // t0 = U1 x0i
      "v_mul_f64 %5, %9, %0\n\t"          // t1 = U1 x0r
      "v_mul_f64 %6, %13, %1\n\t"         // t2 = U5 x0i
      "v_mul_f64 %7, %13, %0\n\t"         // t3 = U5 x0r
      "v_fma_f64 %4, %10, %2, -%4\n\t"    // t0 = U2 x1r - U1 x0i
      "v_fma_f64 %5, %10, %3, %5\n\t"     // t1 = U2 x1i + U1 x0r
      "v_fma_f64 %6, %12, %0, -%6\n\t"    // t2 = U4 x0r - U5 x0i
      "v_fma_f64 %7, %12, %1, %7\n\t"     // t3 = U4 x0i + U5 x0r
      "v_fma_f64 %4, -%11, %3, %4\n\t"    // t0 -= U3 x1i
      "v_fma_f64 %5, %11, %2, %5\n\t"     // t1 += U3 x1r
      "v_fma_f64 %6, -%15, %3, %6\n\t"    // t2 -= U7 x1i
      "v_fma_f64 %7, %15, %2, %7\n\t"     // t3 += U7 x1r
      "v_fma_f64 %0, %8, %0, %4\n\t"      // x0r = U0 x0r + t0
      "v_fma_f64 %1, %8, %1, %5\n\t"      // x0i = U0 x0i + t1
      "v_fma_f64 %2, %14, %2, %6\n\t"     // x1r = U6 x1r + t2
      "v_fma_f64 %3, %14, %3, %7"         // x1i = U6 x1i + t3
      : "+v"(x0r), "+v"(x0i), "+v"(x1r), "+v"(x1i), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
      : "v"(U[0]), "v"(U[1]), "v"(U[2]), "v"(U[3]), "v"(U[4]), "v"(U[5]), "v"(U[6]), "v"(U[7]));
}

// (In kernels_circuit.hip an experiment, BORNVI_U_SGPR; in kernels_circuit8.hip the production form: the matrices come by
// scalar loads straight from the gate array.)
// The same gate with the matrix in SCALAR registers (experiment BORNVI_U_SGPR, tools/probes): every instruction reads
// exactly one matrix element, i.e. one SGPR pair -- inside the constant-bus limit of a VOP3 instruction.
__device__ __forceinline__ void gate_pair_inplace_s(double& x0r, double& x0i, double& x1r, double& x1i,
                                                    const double (&U)[8]) {
  double t0, t1, t2, t3;
  asm("v_mul_f64 %4, %9, %1\n\t"
      "v_mul_f64 %5, %9, %0\n\t"
      "v_mul_f64 %6, %13, %1\n\t"
      "v_mul_f64 %7, %13, %0\n\t"
      "v_fma_f64 %4, %10, %2, -%4\n\t"
      "v_fma_f64 %5, %10, %3, %5\n\t"
      "v_fma_f64 %6, %12, %0, -%6\n\t"
      "v_fma_f64 %7, %12, %1, %7\n\t"
      "v_fma_f64 %4, -%11, %3, %4\n\t"
      "v_fma_f64 %5, %11, %2, %5\n\t"
      "v_fma_f64 %6, -%15, %3, %6\n\t"
      "v_fma_f64 %7, %15, %2, %7\n\t"
      "v_fma_f64 %0, %8, %0, %4\n\t"
      "v_fma_f64 %1, %8, %1, %5\n\t"
      "v_fma_f64 %2, %14, %2, %6\n\t"
      "v_fma_f64 %3, %14, %3, %7"
      : "+v"(x0r), "+v"(x0i), "+v"(x1r), "+v"(x1i), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
      : "s"(U[0]), "s"(U[1]), "s"(U[2]), "s"(U[3]), "s"(U[4]), "s"(U[5]), "s"(U[6]), "s"(U[7]));
}

__device__ __forceinline__ void load_u(const double2* __restrict__ Um, double (&U)[8]) {
  const double2 u00 = Um[0], u01 = Um[1], u10 = Um[2], u11 = Um[3];
  U[0] = u00.x; U[1] = u00.y; U[2] = u01.x; U[3] = u01.y; U[4] = u10.x; U[5] = u10.y; U[6] = u11.x; U[7] = u11.y;
}


// Vector memory ops the COMPILER does not track (it would otherwise drain vmcnt to 0 at every loop merge, i.e.
// wait for the previous tile's stores and for the next tile's loads in the middle of the pipeline).  The kernel
// waits by hand: s_waitcnt vmcnt(N) = all but the wave's N youngest vector-memory ops are done, loads and
// stores counted together in issue order (MI355X_MICROARCH.md, cycle constants).
// The destination is a READ-WRITE operand ("+v"): the load lands in the very register that already carries the
// loop-carried variable.  As a plain output ("=v") the compiler may give the asm a fresh register and COPY it into
// the variable's register right behind the asm statement -- i.e. before the data has arrived (seen on gfx950: the
// second tile of every workgroup was computed from stale registers).
// cache-policy bits of the tile loads / stores (A/B switches: 0 none, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc1 nt).  Every byte of a
// state is read once and written once per pass: non-temporal on both sides.  Same-box A/B (tools/probes/circuit_ab.py),
// batch of n = 16, L = 6 / n = 20, L = 8: none 2.150 / 95.7 ms; loads nt 2.065 / 95.1; stores nt 2.139 / 94.3; both
// 2.080 / 93.9; loads sc1 2.149 / 95.6; loads sc1 nt 2.112 / 95.0; loads nt + stores sc1 2.169 / 94.6.
#ifndef BORNVI_LOAD_POLICY
#define BORNVI_LOAD_POLICY 1
#endif
#ifndef BORNVI_STORE_POLICY
#define BORNVI_STORE_POLICY 1
#endif
#define BORNVI_POLICY_STR_0 ""
#define BORNVI_POLICY_STR_1 " nt"
#define BORNVI_POLICY_STR_2 " sc1"
#define BORNVI_POLICY_STR_3 " sc0 sc1"
#define BORNVI_POLICY_STR_4 " sc1 nt"
#define BORNVI_POLICY_CAT_(X_) BORNVI_POLICY_STR_##X_
#define BORNVI_POLICY_CAT(X_) BORNVI_POLICY_CAT_(X_)
#define BORNVI_LOAD_MOD BORNVI_POLICY_CAT(BORNVI_LOAD_POLICY)
#define BORNVI_STORE_MOD BORNVI_POLICY_CAT(BORNVI_STORE_POLICY)
// the 8-byte probability stores of the last pass: the circuit-ending CNOT ring puts the lanes' parity into a high address
// bit, so one store instruction writes every other 8-byte element of two 512-byte runs and a neighbouring instruction
// of the same wave fills the gaps -- these must meet in the L2 (write-back), not stream past it
#ifndef BORNVI_STORE8_POLICY
#define BORNVI_STORE8_POLICY 0
#endif
#define BORNVI_STORE8_MOD BORNVI_POLICY_CAT(BORNVI_STORE8_POLICY)
__device__ __forceinline__ void async_load16(d2_t& dst, uint32_t byte_off, const void* base) {
  asm volatile("global_load_dwordx4 %0, %1, %2" BORNVI_LOAD_MOD : "+v"(dst) : "v"(byte_off), "s"(base) : "memory");
}
// (the s_nop covers the "VMEM store of more than 64 bits, then VALU write of its data registers" hazard: the
// compiler pads its own stores but does not look inside inline asm, and it reuses the data registers at once)
__device__ __forceinline__ void async_store16(uint32_t byte_off, d2_t val, void* base) {
  asm volatile("global_store_dwordx4 %0, %1, %2" BORNVI_STORE_MOD "\n\ts_nop 1" : : "v"(byte_off), "v"(val), "s"(base) : "memory");
}
__device__ __forceinline__ void async_store8(uint32_t byte_off, double val, void* base) {
  asm volatile("global_store_dwordx2 %0, %1, %2" BORNVI_STORE8_MOD : : "v"(byte_off), "v"(val), "s"(base) : "memory");
}

}  // namespace bornvi
