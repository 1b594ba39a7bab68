// Launchers implemented in kernels_circuit.hip / kernels_stein.hip, called from api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "bornvi.h"

namespace bornvi {

// ---- circuit ------------------------------------------------------------------------------------
// Prefix sharing inside a parameter-shift batch (api.hip: circuit_batch): the circuits of the batch are ordered by
// the first pass their shifted parameter touches, slot 0 holds the base circuit; a pass runs the circuits that
// already differ from the base circuit, and those with index >= fresh_begin read the base circuit's state.
struct PrefixShare {
  long long fresh_begin = (1ll << 62);   // circuits >= fresh_begin read slot 0 of the input buffer
  const int* row_map = nullptr;          // final pass: output row of circuit b (< 0: goes to `trash`); null = row b
  double* trash = nullptr;               // 2^n doubles
};
// shift_tab (or null): per circuit of this launch, parameter * 2 + (1 for the minus shift), < 0 for the base circuit.
// gates: [batch][slots][8] doubles, slots >= nfused + 1.  normalise = 1 (circuit_pass_r3_kernel): pivot-normalised records
// (kernels_circuit.hip: build_gates_kernel) and, in slot nfused, the scale of the circuit's probabilities.
hipError_t launch_build_gates(const uint32_t* plan, int nfused, const double* thetas, long long theta_stride,
                              int shift_mode, int p_begin, int p_stride, int include_base, long long b_offset, int batch,
                              double* gates, const int* shift_tab, int slots, int normalise, hipStream_t st);
hipError_t launch_normalise_gates(double* gates, int count, hipStream_t st);   // raw [count][8] -> records, in place
hipError_t prepare_circuit_kernel(size_t lds_bytes);
hipError_t read_circuit_stamps(unsigned long long* out16);   // diagnostic builds (BORNVI_STAMPS) only: zeros otherwise
hipError_t launch_circuit_pass(const uint32_t* plan, uint32_t pass_off, int n, int k, int threads, size_t lds, int batch,
                               const void* in, void* out, double* probs, const double* gates,
                               long long gate_stride, int max_workgroups, int dbg, hipStream_t st);
// fast path (plan.hpp: build_fast_tables); persistent grid of at most max_workgroups workgroups
int circuit_fast_workgroups_per_cu(int threads, size_t lds);
hipError_t launch_circuit_pass_fast(const uint32_t* plan, uint32_t pass_off, const uint32_t* fast, uint32_t fast_off,
                                    int n, int k, size_t lds, int batch, const void* in, void* out, double* probs,
                                    const double* gates, long long gate_stride, int max_workgroups, size_t lds_tab_off,
                                    size_t lds_mats2_off, int direct_mask, int dbg, const PrefixShare& share, hipStream_t st);
// 8 amplitudes per thread, compact tables (kernels_circuit8.hip; plan.hpp: CompactTables)
hipError_t prepare_circuit_r3_kernel(size_t lds_bytes);
int circuit_r3_workgroups_per_cu(int threads, size_t lds);
hipError_t launch_circuit_pass_r3(const uint32_t* plan, uint32_t pass_off, const uint32_t* ctab, uint32_t ct_off, int n, int k,
                                  size_t lds, int batch, const void* in, void* out, double* probs, const double* gates,
                                  long long gate_stride, int max_workgroups, int direct_mask, const PrefixShare& share,
                                  const double* wdot, double* partials, hipStream_t st);
// grad[p] = s * (sum_g partials[(2p) * tiles + g] - sum_g partials[(2p + 1) * tiles + g]),  s = 1/2 (ksd2 == null) or
// 1/2 / sqrt(max(ksd2, 1e-12)) with the clamp's zero gradient below 1e-12; loss_out (or null) = sqrt(max(ksd2, 1e-12))
hipError_t launch_dot_finish(const double* partials, int n_shift, long long tiles, const double* ksd2, double* grad,
                             double* loss_out, hipStream_t st);
hipError_t launch_gate1q(double* state, int n, long long batch, int wire, const double* U, hipStream_t st);
hipError_t launch_cnot(double* state, int n, long long batch, int control, int target, hipStream_t st);
hipError_t launch_born_probs(const double* state, double* probs, int n, long long batch, hipStream_t st);
hipError_t launch_shift_dot(const double* shifted, int n_shift, const double* w, const double* ksd2, int n,
                            double* grad, double* loss_out, hipStream_t st);
hipError_t launch_clip_cast(const double* g64, int P, double max_norm, float* g32, float* norm_out, const double* loss,
                            float* found_inf, hipStream_t st);
hipError_t launch_clip_adam(const double* g64, int P, double max_norm, const double* loss, float* theta, float* g32,
                            double* theta64, float* exp_avg, float* exp_avg_sq, int* counters, const double* lr_table,
                            int n_lr, double beta1, double beta2, double eps, float* norm_out, double* loss_hist,
                            float* norm_hist, hipStream_t st);
hipError_t launch_dldq(const double* y, const double* ksd2, int n, double* dLdq, double* loss_out, hipStream_t st);

// ---- adjoint differentiation (kernels_adjoint.hip): gate-block walks over one / two states ------------
struct AdjRotBlock {      // consecutive one-qubit gates of one wire, applied e = 0 first (kinds: plan.hpp GateKind)
  int wire, nrot;
  int kind[4];
  int param[4];           // parameter index or -1
};
constexpr int ADJ_MAX_CZ = 136;     // CZ gates of one entangler block (all_to_all at n = 17: 136 pairs)
struct AdjEntangler {     // psi'[y] = s(y) psi[A y]: rows of A over the PHYSICAL index bits, CZ signs on linear functions of y
  unsigned row[32];       // bit b of x = parity(y & row[b])
  int ncz;
  unsigned za[ADJ_MAX_CZ], zb[ADJ_MAX_CZ];
};
int adjoint_workgroups(int n);   // workgroups of a rotation-block launch (partials per block = 4 * this)
hipError_t launch_adj_init(double* state, int n, hipStream_t st);
hipError_t launch_adj_rot_forward(double* state, int n, const AdjRotBlock& blk, const double* theta, hipStream_t st);
hipError_t launch_adj_rot_backward(double* phi, double* lam, int n, const AdjRotBlock& blk, const double* theta, double* partials,
                                   hipStream_t st);
hipError_t launch_adj_entangle(const double* in, double* out, const double* in2, double* out2, int n, const AdjEntangler& E,
                               int sign_on_src, hipStream_t st);
hipError_t launch_adj_lambda(const double* psi, const double* w, double* lam, int n, hipStream_t st);
hipError_t launch_adj_reduce(const double* partials, const int* slot_param, int nslots, int nwg, double* grad, hipStream_t st);

// ---- Stein ----------------------------------------------------------------------------------------
hipError_t launch_score(const bornvi_bn_desc& bn, int n, double* S, double* pxz, hipStream_t st);
// `ld`: row pitch of K in doubles (>= 2^n, even)
hipError_t launch_gram_build(int n, double length_scale, const double* S, double* K, long long row_begin,
                             long long row_end, long long ld, hipStream_t st);
hipError_t launch_kp_pairs(int n, double length_scale, long long M, const long long* zi, const long long* zj,
                           const double* si, const double* sj, double* out, hipStream_t st);
size_t quadform_partials(long long rows);  // number of per-workgroup partial sums
hipError_t launch_quadform(int n, const double* K, long long row_begin, long long row_end, const double* q,
                           double* y_or_null, double* ksd2, double* partials, hipStream_t st);
// batched Y = K Q^T on the matrix cores (kernels_batched.hip): n >= 8, B >= 2; Y must be given ([B, 2^n])
bool quadform_batched_supported(int n, int B);
hipError_t launch_quadform_batched(int n, const double* K, long long ld, const double* Q, int B, double* Y, double* ksd2, hipStream_t st);
size_t quadform_sym_workspace_doubles(int n);
hipError_t launch_quadform_sym(int n, const double* K, long long ld, const double* q, double* y_or_null, double* ksd2,
                               double* ws, hipStream_t st);
hipError_t launch_quadform_sym_pairs(int n, const double* K_lo, const double* K_hi, long long ld, long long pair_begin,
                                     long long pair_end, const double* q, double* y_or_null, double* ksd2, double* ws,
                                     hipStream_t st);
int quadform_sym_rows_per_strip();
int quadform_sym_min_n();     // below it the symmetric entry point runs the full-matrix kernel, and there are no strip pairs
// matrix-free mat-vec helpers
hipError_t launch_kron_pack(int n, double length_scale, const double* S, const double* q,
                            double* packed /*complex [(n+2)/2, 2^n]*/, double* gate /*[8]*/, hipStream_t st);
hipError_t launch_kron_combine(int n, const double* S, const double* q, const double* packed, double* y,
                               double* partials, double* ksd2, hipStream_t st);
size_t kron_partials(int n);

}  // namespace bornvi
