// Adjoint differentiation of the Born-machine circuits (gfx950) -- the OPT-IN second gradient engine of SURVEY.md
// section 8(f) row 4.  The reference differentiates with diff_method="parameter-shift" (quantum_born_machine.py:58, :90,
// :114): 2P circuit evaluations per gradient.  For L(theta) = f(q), q_z = |psi_z|^2 and w = dL/dq,
//     dL/dtheta_k = 2 Re <lambda_k | dU_k/dtheta_k | phi_{k-1}> = Im <lambda_k | P_k | phi_k>,   U_k = exp(-i theta_k P_k / 2),
// with phi_k the state after gate k and lambda_k = U_{k+1}^+ ... U_G^+ (w o psi): one forward walk and one backward walk
// over the gate list with two states -- about three circuit evaluations instead of 2P.  Same gradient to rounding;
// it changes what "2P evaluations" means, so the trainer uses it only when asked (grad_engine = "adjoint").
//
// The walk is gate-block by gate-block, each block one HBM round trip over the state(s):
//   rotation block   the consecutive one-qubit gates of one wire (H | RX RY RZ | RY RZ): pairs of amplitudes in
//                    registers; backward it also accumulates Im <lambda| P |phi> for each of its parameters
//                    (per-workgroup partials, fixed order: deterministic) before undoing each rotation on both states;
//   entangler block  the consecutive CNOT / CZ gates between two rotation layers: psi'[y] = s(y) psi[A y] with a
//                    GF(2)-linear index map A (product of the CNOT maps) and the CZ signs evaluated on linear
//                    functions of y -- one gather per block instead of one pass per gate.
#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "plan.hpp"

namespace bornvi {

namespace {

__device__ __forceinline__ double wave_sum_adj(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

struct C2 { double re, im; };
__device__ __forceinline__ C2 cmul(C2 a, C2 b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }

// one-qubit gate of kind `kind` with angle t applied to the pair (x0, x1); inverse = true applies the adjoint
__device__ __forceinline__ void apply_1q(int kind, double c, double s, bool inverse, C2& x0, C2& x1) {
  if (kind == G_H) {
    const double h = 0.70710678118654752440;
    const C2 a = x0, b = x1;
    x0 = {h * (a.re + b.re), h * (a.im + b.im)};
    x1 = {h * (a.re - b.re), h * (a.im - b.im)};
    return;
  }
  if (inverse) s = -s;           // R(t)^+ = R(-t)
  const C2 a = x0, b = x1;
  if (kind == G_RX) {            // [[c, -i s], [-i s, c]]
    x0 = {c * a.re + s * b.im, c * a.im - s * b.re};
    x1 = {s * a.im + c * b.re, -s * a.re + c * b.im};
  } else if (kind == G_RY) {     // [[c, -s], [s, c]]
    x0 = {c * a.re - s * b.re, c * a.im - s * b.im};
    x1 = {s * a.re + c * b.re, s * a.im + c * b.im};
  } else {                       // RZ: diag(e^{-i t/2}, e^{+i t/2})
    x0 = cmul({c, -s}, a);
    x1 = cmul({c, s}, b);
  }
}

// Im <l | P | a> summed over the pair, P the Pauli generator of `kind`
__device__ __forceinline__ double im_lpa(int kind, C2 l0, C2 l1, C2 a0, C2 a1) {
  C2 v0, v1;
  if (kind == G_RX) { v0 = a1; v1 = a0; }
  else if (kind == G_RY) { v0 = {a1.im, -a1.re}; v1 = {-a0.im, a0.re}; }     // -i a1, i a0
  else { v0 = a0; v1 = {-a1.re, -a1.im}; }
  return (l0.re * v0.im - l0.im * v0.re) + (l1.re * v1.im - l1.im * v1.re);
}

constexpr int ADJ_THREADS = 256;

// forward: state <- U_block state
__global__ __launch_bounds__(ADJ_THREADS) void adj_rot_forward_kernel(double2* __restrict__ state, long long npairs, int pbit,
                                                                      AdjRotBlock blk, const double* __restrict__ theta) {
  double c[4], s[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    c[e] = 1.0; s[e] = 0.0;
    if (e < blk.nrot && blk.param[e] >= 0) sincos(0.5 * theta[blk.param[e]], &s[e], &c[e]);
  }
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long lowmask = (1ll << pbit) - 1;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < npairs; idx += stride) {
    const long long i0 = ((idx & ~lowmask) << 1) | (idx & lowmask);
    const long long i1 = i0 | (1ll << pbit);
    const double2 v0 = state[i0], v1 = state[i1];
    C2 a0 = {v0.x, v0.y}, a1 = {v1.x, v1.y};
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (e < blk.nrot) apply_1q(blk.kind[e], c[e], s[e], false, a0, a1);
    state[i0] = make_double2(a0.re, a0.im);
    state[i1] = make_double2(a1.re, a1.im);
  }
}

// backward: for e = last .. first: g_e += Im <lambda| P_e |phi>;  phi <- U_e^+ phi;  lambda <- U_e^+ lambda.
// partials[(blockIdx.x) * 4 + e] = this workgroup's share of g_e.
__global__ __launch_bounds__(ADJ_THREADS) void adj_rot_backward_kernel(double2* __restrict__ phi, double2* __restrict__ lam,
                                                                       long long npairs, int pbit, AdjRotBlock blk,
                                                                       const double* __restrict__ theta, double* __restrict__ partials) {
  __shared__ double red[ADJ_THREADS / 64][4];
  double c[4], s[4], g[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    c[e] = 1.0; s[e] = 0.0; g[e] = 0.0;
    if (e < blk.nrot && blk.param[e] >= 0) sincos(0.5 * theta[blk.param[e]], &s[e], &c[e]);
  }
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long lowmask = (1ll << pbit) - 1;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < npairs; idx += stride) {
    const long long i0 = ((idx & ~lowmask) << 1) | (idx & lowmask);
    const long long i1 = i0 | (1ll << pbit);
    const double2 p0 = phi[i0], p1 = phi[i1], q0 = lam[i0], q1 = lam[i1];
    C2 a0 = {p0.x, p0.y}, a1 = {p1.x, p1.y}, l0 = {q0.x, q0.y}, l1 = {q1.x, q1.y};
#pragma unroll
    for (int e = 3; e >= 0; --e)
      if (e < blk.nrot) {
        if (blk.param[e] >= 0) g[e] += im_lpa(blk.kind[e], l0, l1, a0, a1);
        apply_1q(blk.kind[e], c[e], s[e], true, a0, a1);
        apply_1q(blk.kind[e], c[e], s[e], true, l0, l1);
      }
    phi[i0] = make_double2(a0.re, a0.im); phi[i1] = make_double2(a1.re, a1.im);
    lam[i0] = make_double2(l0.re, l0.im); lam[i1] = make_double2(l1.re, l1.im);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const double v = wave_sum_adj(g[e]);
    if (lane == 0) red[wave][e] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    double v = 0.0;
    for (int w = 0; w < ADJ_THREADS / 64; ++w) v += red[w][threadIdx.x];     // fixed order
    partials[(long long)blockIdx.x * 4 + threadIdx.x] = v;
  }
}

// out[y] = s(e) in[x],  x = M y (GF(2)),  e = y (sign_on_src = 0: forward) or x (sign_on_src = 1: backward);
// s(e) = (-1)^{sum_k parity(e & za[k]) parity(e & zb[k])}.  Two states at once when in2 != nullptr.
__global__ __launch_bounds__(ADJ_THREADS) void adj_entangle_kernel(const double2* __restrict__ in, double2* __restrict__ out,
                                                                   const double2* __restrict__ in2, double2* __restrict__ out2,
                                                                   long long N, int n, AdjEntangler E, int sign_on_src) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long y = (long long)blockIdx.x * blockDim.x + threadIdx.x; y < N; y += stride) {
    unsigned x = 0;
    for (int b = 0; b < n; ++b) x |= (unsigned)(__popc((unsigned)y & E.row[b]) & 1) << b;
    const unsigned e = sign_on_src ? x : (unsigned)y;
    unsigned sg = 0;
    for (int k = 0; k < E.ncz; ++k) sg ^= (unsigned)(__popc(e & E.za[k]) & __popc(e & E.zb[k]) & 1);
    const double f = sg ? -1.0 : 1.0;
    const double2 v = in[x];
    out[y] = make_double2(f * v.x, f * v.y);
    if (in2) {
      const double2 u = in2[x];
      out2[y] = make_double2(f * u.x, f * u.y);
    }
  }
}

// lambda = w o psi (w real: dL/dq)
__global__ __launch_bounds__(ADJ_THREADS) void adj_lambda_kernel(const double2* __restrict__ psi, const double* __restrict__ w,
                                                                 double2* __restrict__ lam, long long N) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) {
    const double2 v = psi[i];
    const double f = w[i];
    lam[i] = make_double2(f * v.x, f * v.y);
  }
}

__global__ void adj_init_kernel(double2* __restrict__ state, long long N) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride)
    state[i] = make_double2(i == 0 ? 1.0 : 0.0, 0.0);
}

// grad[param] = sum over the workgroups of a rotation block of its partial (fixed order); one thread per (block, slot)
__global__ void adj_reduce_kernel(const double* __restrict__ partials, const int* __restrict__ slot_param, int nslots, int nwg,
                                  double* __restrict__ grad) {
  const int sidx = blockIdx.x * blockDim.x + threadIdx.x;
  if (sidx >= nslots) return;
  const int p = slot_param[sidx];
  if (p < 0) return;
  const int blk = sidx >> 2, e = sidx & 3;
  double v = 0.0;
  for (int w = 0; w < nwg; ++w) v += partials[((long long)blk * nwg + w) * 4 + e];
  grad[p] = v;
}

unsigned adj_grid(long long work) {
  long long g = (work + ADJ_THREADS - 1) / ADJ_THREADS;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (unsigned)g;
}
}  // namespace

int adjoint_workgroups(int n) { return (int)adj_grid(1ll << (n > 0 ? n - 1 : 0)); }

hipError_t launch_adj_init(double* state, int n, hipStream_t st) {
  adj_init_kernel<<<adj_grid(1ll << n), ADJ_THREADS, 0, st>>>((double2*)state, 1ll << n);
  return hipGetLastError();
}

hipError_t launch_adj_rot_forward(double* state, int n, const AdjRotBlock& blk, const double* theta, hipStream_t st) {
  const long long npairs = 1ll << (n - 1);
  adj_rot_forward_kernel<<<adj_grid(npairs), ADJ_THREADS, 0, st>>>((double2*)state, npairs, n - 1 - blk.wire, blk, theta);
  return hipGetLastError();
}

hipError_t launch_adj_rot_backward(double* phi, double* lam, int n, const AdjRotBlock& blk, const double* theta, double* partials,
                                   hipStream_t st) {
  const long long npairs = 1ll << (n - 1);
  adj_rot_backward_kernel<<<adj_grid(npairs), ADJ_THREADS, 0, st>>>((double2*)phi, (double2*)lam, npairs, n - 1 - blk.wire, blk, theta,
                                                                   partials);
  return hipGetLastError();
}

hipError_t launch_adj_entangle(const double* in, double* out, const double* in2, double* out2, int n, const AdjEntangler& E,
                               int sign_on_src, hipStream_t st) {
  adj_entangle_kernel<<<adj_grid(1ll << n), ADJ_THREADS, 0, st>>>((const double2*)in, (double2*)out, (const double2*)in2, (double2*)out2,
                                                                1ll << n, n, E, sign_on_src);
  return hipGetLastError();
}

hipError_t launch_adj_lambda(const double* psi, const double* w, double* lam, int n, hipStream_t st) {
  adj_lambda_kernel<<<adj_grid(1ll << n), ADJ_THREADS, 0, st>>>((const double2*)psi, w, (double2*)lam, 1ll << n);
  return hipGetLastError();
}

hipError_t launch_adj_reduce(const double* partials, const int* slot_param, int nslots, int nwg, double* grad, hipStream_t st) {
  if (nslots <= 0) return hipSuccess;
  adj_reduce_kernel<<<(unsigned)((nslots + 255) / 256), 256, 0, st>>>(partials, slot_param, nslots, nwg, grad);
  return hipGetLastError();
}

}  // namespace bornvi
