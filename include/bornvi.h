/*
 * bornvi.h -- C ABI of libbornvi_hip.so, the MI355X (gfx950) backend for the KSD
 * variational-inference hot path of sozoluffy/TensorNetworks.
 *
 * The reference has no FFI: its boundary for this path is the Python API of
 * quantum_born_machine.py / stein_utils.py / ksd_vi_quantum.py.  Each entry point below
 * states which reference lines it replaces; INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - plain C, no C++ types, no exceptions across the boundary;
 *   - return value: 0 (BORNVI_OK) or a negative bornvi_status; text via bornvi_last_error();
 *   - every pointer documented "dev" is device memory owned by the CALLER (PyTorch);
 *     the library never frees it and keeps no reference after the call returns;
 *   - every op is asynchronous on the hipStream_t passed as `stream` (0 = null stream);
 *   - workspace is caller-provided; bornvi_*_workspace_bytes gives the size for the full
 *     batch, a smaller workspace makes the library process the batch in chunks
 *     (it must hold at least one circuit);
 *   - a handle is bound to one device and is not thread-safe; distinct handles are.
 *   - outcome index i <-> bitstring bin(i).zfill(n): bit position 0 of the tuple is the MOST
 *     significant bit of i and is wire 0 of the circuit (utils.py:77-91).
 */
#ifndef BORNVI_H
#define BORNVI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: exactly the functions declared in this header are exported. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define BORNVI_VERSION 100 /* 0.1.0 */

typedef struct bornvi_ctx* bornvi_handle;
typedef void* bornvi_stream; /* hipStream_t */

typedef enum {
  BORNVI_OK = 0,
  BORNVI_ERR_INVALID = -1,     /* bad argument */
  BORNVI_ERR_HIP = -2,         /* a HIP runtime call failed (text has the HIP error) */
  BORNVI_ERR_WORKSPACE = -3,   /* workspace too small for a single circuit */
  BORNVI_ERR_UNSUPPORTED = -4  /* size outside the supported range */
} bornvi_status;

/* ansatz_type strings of QuantumBornMachine (quantum_born_machine.py:31-38, :57-128) */
typedef enum {
  BORNVI_ANSATZ_HARDWARE_EFFICIENT = 0, /* quantum_born_machine.py:58-87  */
  BORNVI_ANSATZ_ALL_TO_ALL = 1,         /* quantum_born_machine.py:90-111 */
  BORNVI_ANSATZ_BASIC = 2               /* quantum_born_machine.py:114-128 */
} bornvi_ansatz;

int bornvi_version(void);

/* One handle per device.  Holds the compiled circuit plans (small device buffers). */
int bornvi_create(int device_ordinal, bornvi_handle* out);
void bornvi_destroy(bornvi_handle h);
/* Text of the last error on this handle (valid until the next call on it). h may be NULL
 * for errors of bornvi_create. */
const char* bornvi_last_error(bornvi_handle h);

/* A HIP stream restricted to the compute units [first_cu, first_cu + num_cus) of the device
 * (hipExtStreamCreateWithCUMask): lets the instruction-bound circuit passes and the HBM-bound contraction of one
 * step run side by side on disjoint halves of the chip instead of taking turns.  Launch the
 * circuit entry points on such a stream with the option "circuit_cus" = num_cus so that the persistent grid is
 * sized for it.  No reference counterpart. */
int bornvi_stream_create_cu_range(bornvi_handle h, int first_cu, int num_cus, bornvi_stream* out);
int bornvi_stream_destroy(bornvi_handle h, bornvi_stream stream);

/* Tuning knobs of the circuit planner (clears the plan cache): "tile_bits" (4..13, amplitudes per
 * LDS tile = 2^tile_bits, for every n), "tile_bits_multi" (tile size used only when the state needs
 * several tiles; default 0 = chosen per plan: 13 wherever the persistent kernel can run such tiles, else 11), "low_bits" (0..8, contiguous 16-byte elements per HBM run =
 * 2^low_bits), "max_threads" (64..1024); "debug_flags" (timing-only ablations of the circuit kernel:
 * results are INVALID while non-zero).  Engine switches (no effect on results): "fast_path",
 * "fast_workgroups_per_cu", "workgroups_per_cu", "direct_stages", "circuit_cus", "zero_support" (default 1: the first two passes of a circuit from |0..0> leave out what the support of that state makes
 * known zeros -- tiles nobody reads are not written, slots known to be zero are not loaded), "reg_wires" (default 3: plans with 8 amplitudes per thread for the pass kernel that holds four waves per SIMD, wherever
 * such a plan is eligible; 4: 16 amplitudes per thread, two waves per SIMD -- also the fallback; clears the plan cache),
 * "read_map" (default -1 = by the kernel: the planner may fold phase-0 CNOTs on thread-held wires into a stage's read map --
 * fewer stages, but such a stage needs a barrier between its reads and its write-back: faster with 8 amplitudes per
 * thread, slower with 16; 0 / 1 force it; clears the plan cache), "contig_out" (default 1: the buffer a pass writes keeps that
 * pass's own local wires on its low address bits -- a tile is stored as one contiguous block, the next pass reads runs; 0: both
 * sides move 1-KiB runs; clears the plan cache), "alternate_walk" (default 1: odd passes
 * walk the tiles of the batch from the last to the first, so that a pass starts on the states the previous one wrote last --
 * the ones the memory-side cache still holds), "batched_quadform" (0: bornvi_stein_quadform
 * with B > 1 runs B GEMV passes instead of one matrix-core pass); "grad_engine" (default 0 = the reference's
 * parameter-shift rule; 1 = OPT-IN: bornvi_paramshift_grad answers with adjoint differentiation, see below);
 * "prefix_share" (default 0; 1: in
 * bornvi_paramshift_probs* a shifted circuit starts from the base circuit's state at the first pass
 * its parameter touches instead of |0..0> -- the rows are bit-identical, fewer passes are run). */
int bornvi_set_option(bornvi_handle h, const char* name, long long value);
/* Current value of a planner / engine option ("reg_wires": 3 = the pass kernel with 8 amplitudes per thread and four waves
 * per SIMD where the plan is eligible, 4 = 16 per thread; "read_map", "contig_out", "tile_bits", "tile_bits_multi", "low_bits",
 * "prefix_share", "grad_engine", "fast_path", "direct_stages", "zero_support", "alternate_walk", "batched_quadform"). */
int bornvi_get_option(bornvi_handle h, const char* name, long long* value);

/* num_ansatz_params (quantum_born_machine.py:31-38) and gate count of the QNode. */
int bornvi_num_params(int ansatz, int n, int layers);
int bornvi_num_gates(int ansatz, int n, int layers);

/* ---- circuit -> Born probabilities (replaces the PennyLane QNode `pqc`, -------------------
 * quantum_born_machine.py:58-128, as called by get_probabilities :132-137).
 * thetas dev [batch, P] float64;  probs dev [batch, 2^n] float64, lexicographic, wire 0 = MSB. */
size_t bornvi_circuit_workspace_bytes(bornvi_handle h, int ansatz, int n, int layers, int batch);
int bornvi_circuit_probs(bornvi_handle h, int ansatz, int n, int layers, int batch,
                         const double* thetas, double* probs,
                         void* workspace, size_t workspace_bytes, bornvi_stream stream);

/* ---- parameter-shift evaluations (replaces diff_method="parameter-shift", ------------------
 * quantum_born_machine.py:58,:90,:114, triggered by loss.backward() ksd_vi_quantum.py:150).
 * For p in [p_begin, p_end): circuits theta + pi/2 e_p and theta - pi/2 e_p.
 * probs dev [(include_base ? 1 : 0) + 2 (p_end - p_begin), 2^n]: optional base row first,
 * then rows (+p_begin, -p_begin, +p_begin+1, -p_begin+1, ...).  theta dev [P] float64.
 * The batch for bornvi_circuit_workspace_bytes is that row count. */
int bornvi_paramshift_probs(bornvi_handle h, int ansatz, int n, int layers,
                            const double* theta, int p_begin, int p_end, int include_base,
                            double* probs, void* workspace, size_t workspace_bytes,
                            bornvi_stream stream);
/* The same for the parameters p_begin, p_begin + p_stride, ... (p_count of them): rows (+p_i, -p_i) in that order.
 * One rank of W takes p_begin = rank, p_stride = W: with prefix sharing a parameter of an early layer costs more
 * passes than one of a late layer, and the interleaved deal keeps the ranks level (no reference counterpart: the
 * reference evaluates all shifts in one process). */
int bornvi_paramshift_probs_strided(bornvi_handle h, int ansatz, int n, int layers,
                                    const double* theta, int p_begin, int p_count, int p_stride,
                                    int include_base, double* probs,
                                    void* workspace, size_t workspace_bytes, bornvi_stream stream);

/* grad[p - p_begin] = 1/2 * sum_z dLdq[z] (q(theta + pi/2 e_p)[z] - q(theta - pi/2 e_p)[z]).
 * dLdq dev [2^n], grad dev [p_end - p_begin].  Workspace: the circuit workspace for batch
 * 2 (p_end - p_begin) plus 2 (p_end - p_begin) * 2^n * 8 bytes (bornvi_paramshift_grad_workspace_bytes). */
size_t bornvi_paramshift_grad_workspace_bytes(bornvi_handle h, int ansatz, int n, int layers,
                                              int p_begin, int p_end);
int bornvi_paramshift_grad(bornvi_handle h, int ansatz, int n, int layers,
                           const double* theta, const double* dLdq, int p_begin, int p_end,
                           double* grad, void* workspace, size_t workspace_bytes,
                           bornvi_stream stream);

/* ---- parameter-shift gradient with the dot product FUSED into the last circuit pass (replaces, like
 * bornvi_paramshift_grad, the backward of ksd_vi_quantum.py:150 through diff_method="parameter-shift",
 * quantum_born_machine.py:58): the shifted circuits' probabilities are never written; their last pass accumulates
 * sum_z w_z q(z) per circuit instead (fixed summation order: deterministic).  Two calls around the caller's contraction:
 *   begin : all passes but the last for the base circuit and the 2 p_count shifted ones (parameters p_begin,
 *           p_begin + p_stride, ...), then the base circuit's last pass -> q_out dev [2^n];
 *   (the caller computes y = K_p q, ksd2 = q.y -- any bornvi_stein_* contraction, all-reduced over the ranks)
 *   finish: the shifted circuits' last pass with w = y -> grad dev [p_count]:
 *           grad_p = s * sum_z w_z (q+_p(z) - q-_p(z)),  s = 1/2 if ksd2 == NULL, else 1/2 / sqrt(max(*ksd2, 1e-12)) with the
 *           clamp's zero gradient below 1e-12 (ksd_vi_quantum.py:145); loss_out (or NULL) = sqrt(max(*ksd2, 1e-12)).
 * The SAME workspace (bornvi_paramshift_dot_workspace_bytes, 16-byte aligned) must be passed to both and left alone in
 * between.  Available for multi-pass plans of the 8-amplitude kernel ("reg_wires" = 3) without prefix sharing: the size
 * query returns 0 otherwise (use bornvi_paramshift_probs + bornvi_ksd_grad_finish then). */
size_t bornvi_paramshift_dot_workspace_bytes(bornvi_handle h, int ansatz, int n, int layers, int p_count);
int bornvi_paramshift_dot_begin(bornvi_handle h, int ansatz, int n, int layers, const double* theta, int p_begin,
                                int p_count, int p_stride, double* q_out, void* workspace, size_t workspace_bytes,
                                bornvi_stream stream);
int bornvi_paramshift_dot_finish(bornvi_handle h, int ansatz, int n, int layers, int p_count, const double* w,
                                 const double* ksd2, double* grad, double* loss_out, void* workspace,
                                 size_t workspace_bytes, bornvi_stream stream);

/* ---- un-fused gate application on a batch of statevectors (the gate-apply micro-benchmark of
 * BASELINE.json; one HBM round trip per gate = 32 * 2^n bytes per state).
 * state dev [batch, 2^n] complex128 (re, im interleaved), updated in place.
 * U: HOST pointer to 8 doubles (u00.re, u00.im, u01.re, u01.im, u10.re, ..., u11.im). */
int bornvi_gate1q_apply(bornvi_handle h, int n, long long batch, double* state, int wire,
                        const double* U, bornvi_stream stream);
int bornvi_cnot_apply(bornvi_handle h, int n, long long batch, double* state, int control,
                      int target, bornvi_stream stream);
/* q = |psi|^2 : state dev [batch, 2^n] complex128 -> probs dev [batch, 2^n] float64. */
int bornvi_born_probs(bornvi_handle h, int n, long long batch, const double* state, double* probs,
                      bornvi_stream stream);

/* ---- score function (replaces stein_utils.compute_prob_joint_xz :58-112 and ----------------
 * get_score_function_sp_for_z :115-136 evaluated for all 2^n latent states, i.e.
 * KSDVariationalInference._precompute_all_s_p ksd_vi_quantum.py:70-75, on top of
 * BayesianNetwork.get_joint_probability bayesian_network.py:111-146).
 * All pointers in the descriptor are DEVICE pointers (layout: pack_network() in
 * tensornetworks_amd/bayesian_network.py). */
typedef struct {
  int32_t num_nodes;        /* V <= 64 */
  int32_t max_parents;      /* row length of `parents` (8) */
  const int32_t* role;      /* [V] latent position 0..n-1 | -1 observed=0 | -2 observed=1 | -3 summed out */
  const int32_t* n_parents; /* [V] */
  const int32_t* parents;   /* [V, max_parents] node indices, CPT key order */
  const int32_t* cpt_off;   /* [V] offset in doubles into cpt */
  const double* cpt;        /* table[config][value]; config = parent values, first parent MSB */
} bornvi_bn_desc;
/* S dev [2^n, n] (row z, column b = tuple position b), pxz dev [2^n] or NULL. */
int bornvi_score_from_cpts(bornvi_handle h, const bornvi_bn_desc* bn, int n, double* S,
                           double* pxz, bornvi_stream stream);

/* ---- Stein-kernel Gram matrix (replaces the N^2 calls of get_stein_kernel_kp_value, --------
 * stein_utils.py:138-197 with base_hamming_kernel_torch :30-55, made in
 * ksd_vi_quantum.py:125-141).  K dev [2^n, 2^n] float64 row-major. */
int bornvi_stein_gram_build(bornvi_handle h, int n, double length_scale, const double* S,
                            double* K, bornvi_stream stream);

/* Row block [row_begin, row_end) of K_p, K_rows dev [row_end - row_begin, 2^n]: lets each GPU of a
 * node hold 1/W of the Gram matrix (multi-GPU row shard of the quadratic form). */
int bornvi_stein_gram_build_rows(bornvi_handle h, int n, double length_scale, const double* S,
                                 long long row_begin, long long row_end, double* K_rows,
                                 bornvi_stream stream);

/* Padded row pitch.  K_p with a power-of-two row pitch puts the same column of every row into the same HBM channel and
 * bank, and the symmetric contraction streams 32 rows per wave at the same column; bornvi_stein_gram_ld(n) is the
 * pitch (in doubles, >= 2^n, even) the library recommends: 2^n + 32 for n >= 14, else 2^n (n = 16: 2.56 ms on every
 * allocation against 2.62 ... 2.83 ms dense; 0 for n outside [1, 17]).  The `_ld` entry points take any even pitch in
 * [2^n, 2^n + 4096]; columns >= 2^n of a row are never read or written. */
long long bornvi_stein_gram_ld(int n);
int bornvi_stein_gram_build_rows_ld(bornvi_handle h, int n, double length_scale, const double* S,
                                    long long row_begin, long long row_end, double* K_rows, long long ld,
                                    bornvi_stream stream);

/* k_p(z_i, z_j | x) for M explicit pairs -- the batched form of ONE call of
 * stein_utils.get_stein_kernel_kp_value (:138-197): zi, zj dev int64 [M] outcome indices,
 * si, sj dev [M, n] the score rows sp_at_z1 / sp_at_z2 supplied by the caller, out dev [M]. */
int bornvi_stein_kp_pairs(bornvi_handle h, int n, double length_scale, long long M,
                          const long long* zi, const long long* zj, const double* si,
                          const double* sj, double* out, bornvi_stream stream);

/* ---- quadratic form (replaces the accumulation sum_ij q_i q_j k_p, ksd_vi_quantum.py:123-142).
 * Q dev [B, 2^n]; ksd2 dev [B] = q_b^T K q_b; Y dev [B, 2^n] = K q_b or NULL.
 * Deterministic (no atomics).  Workspace: bornvi_stein_quadform_workspace_bytes. */
size_t bornvi_stein_quadform_workspace_bytes(bornvi_handle h, int n, int B);
int bornvi_stein_quadform(bornvi_handle h, int n, const double* K, const double* Q, int B,
                          double* ksd2, double* Y, void* workspace, size_t workspace_bytes,
                          bornvi_stream stream);
/* The same for a K with row pitch `ld` doubles (bornvi_stein_gram_build_rows_ld; even, 2^n <= ld <= 2^n + 4096): the
 * trainer's padded K_p serves the batched contraction too (no second 32 GiB copy).  With ld > 2^n only the batched
 * matrix-core form exists (n >= 8, B >= 2, option "batched_quadform" = 1), otherwise BORNVI_ERR_UNSUPPORTED; B = 1 on
 * a padded K: bornvi_stein_quadform_sym_ld. */
int bornvi_stein_quadform_ld(bornvi_handle h, int n, const double* K, long long ld, const double* Q, int B,
                             double* ksd2, double* Y, void* workspace, size_t workspace_bytes,
                             bornvi_stream stream);

/* Same result for a SYMMETRIC K (every K from bornvi_stein_gram_build is bitwise symmetric): reads
 * only the upper triangle, i.e. half the HBM traffic of bornvi_stein_quadform; deterministic (column
 * partials in the workspace instead of atomics).  q, y dev [2^n]; ksd2 dev [1].  A dense (unpadded) matrix with
 * n < 14 is handed to the full-matrix kernel: too few 256-row bands to fill the chip, and the matrix is cache-sized
 * (n = 12: 33 us against 93 us); same K q to rounding. */
size_t bornvi_stein_quadform_sym_workspace_bytes(bornvi_handle h, int n);
int bornvi_stein_quadform_sym(bornvi_handle h, int n, const double* K, const double* q,
                              double* ksd2, double* y, void* workspace, size_t workspace_bytes,
                              bornvi_stream stream);

/* The same for a K with row pitch `ld` doubles (bornvi_stein_gram_build_rows_ld).  Matrices of fewer than two 256-row
 * bands (n < 9) run the full-matrix kernel, which takes dense rows only: ld > 2^n there is BORNVI_ERR_UNSUPPORTED. */
int bornvi_stein_quadform_sym_ld(bornvi_handle h, int n, const double* K, long long ld, const double* q,
                                 double* ksd2, double* y, void* workspace, size_t workspace_bytes,
                                 bornvi_stream stream);

/* Strip-pair shard of the symmetric form (several GPUs, each reading only its part of the UPPER triangle); n >= 9.
 * The rows are cut into strips ("bands": one workgroup each) of bornvi_stein_sym_strip_rows() = 256 rows; pair p =
 * strips p and n_strips-1-p (a long and a short part of the triangle).  A GPU owning pairs [pair_begin, pair_end) holds K_lo = rows of the strips
 * [pair_begin, pair_end) and K_hi = rows of the strips [n_strips - pair_end, n_strips - pair_begin) (each block
 * built with bornvi_stein_gram_build_rows) and gets y_partial dev [2^n] and ksd2_partial dev [1]: its additive
 * share of K q and of q^T K q; the sum over the GPUs (one all-reduce of 2^n + 1 doubles) is the full result.
 * Workspace as bornvi_stein_quadform_sym. */
int bornvi_stein_sym_strip_rows(void);
int bornvi_stein_quadform_sym_pairs(bornvi_handle h, int n, const double* K_lo, const double* K_hi,
                                    long long pair_begin, long long pair_end, const double* q,
                                    double* ksd2_partial, double* y_partial, void* workspace,
                                    size_t workspace_bytes, bornvi_stream stream);

int bornvi_stein_quadform_sym_pairs_ld(bornvi_handle h, int n, const double* K_lo, const double* K_hi, long long ld,
                                       long long pair_begin, long long pair_end, const double* q,
                                       double* ksd2_partial, double* y_partial, void* workspace,
                                       size_t workspace_bytes, bornvi_stream stream);

/* Row-sharded form: K_rows holds rows [row_begin, row_end); q dev [2^n] (full);
 * y_rows dev [row_end - row_begin] = those rows of K q (or NULL); ksd2_partial dev [1] =
 * sum over them of q_i y_i.  Workspace as bornvi_stein_quadform. */
int bornvi_stein_quadform_rows(bornvi_handle h, int n, const double* K_rows, long long row_begin,
                               long long row_end, const double* q, double* y_rows,
                               double* ksd2_partial, void* workspace, size_t workspace_bytes,
                               bornvi_stream stream);

/* Matrix-free y = K_p q via the Kronecker structure of K_p (needed where the dense Gram does
 * not fit: n = 20 would be 8 TiB).  q, y dev [2^n]; ksd2 dev [1]. */
size_t bornvi_stein_matvec_kron_workspace_bytes(bornvi_handle h, int n);
int bornvi_stein_matvec_kron(bornvi_handle h, int n, double length_scale, const double* S,
                             const double* q, double* y, double* ksd2, void* workspace,
                             size_t workspace_bytes, bornvi_stream stream);

/* ---- loss and gradient assembly (replaces ksd_vi_quantum.py:144-145 and the autograd chain
 * of :150): loss = sqrt(max(ksd2, 1e-12)); dLdq = y / loss (0 when the clamp is active);
 * grad[p] = 1/2 dLdq . (q+_p - q-_p).
 * shifted dev [2 * n_shift, 2^n] rows (+p, -p) as bornvi_paramshift_probs lays them out;
 * y dev [2^n]; ksd2 dev [1]; loss_out dev [1]; dLdq_out dev [2^n] or NULL; grad dev [n_shift]. */
int bornvi_ksd_grad_finish(bornvi_handle h, int n, const double* shifted, int n_shift,
                           const double* y, const double* ksd2, double* loss_out,
                           double* dLdq_out, double* grad, bornvi_stream stream);

/* Gradient hand-off to the optimiser (replaces the float cast of the parameter-shift VJP and
 * torch.nn.utils.clip_grad_norm_(params, gradient_clip_norm), ksd_vi_quantum.py:153): grad32 dev [P] float32 =
 * float32(grad64) * min(1, max_norm / (||float32(grad64)||_2 + 1e-6)); total_norm dev [1] float32 = that norm. */
int bornvi_clip_cast_grad(bornvi_handle h, int P, const double* grad64, double max_norm,
                          float* grad32, float* total_norm, bornvi_stream stream);
/* The same plus the reference's NaN/Inf guard on the loss (ksd_vi_quantum.py:147-148, "Skipping update") as a device
 * flag: found_inf dev [1] float32 = 1 if loss (dev [1] float64) is NaN or +-Inf, else 0 -- the form torch's fused
 * optimisers consume, so a step needs no host read-back of the loss. */
int bornvi_clip_cast_grad_guard(bornvi_handle h, int P, const double* grad64, double max_norm,
                                const double* loss, float* grad32, float* total_norm,
                                float* found_inf, bornvi_stream stream);

/* The whole optimiser hand-off of one epoch in one launch, for the latency-bound sizes (a HIP-graph replay of the step
 * spends most of its nodes in the optimiser otherwise): the clip and guard above, then the update of
 * optim.Adam(lr, betas) (ksd_vi_quantum.py:92-99; eps as given, no weight decay, no amsgrad) on the float32 theta:
 *   m = beta1 m + (1 - beta1) g;  v = beta2 v + (1 - beta2) g^2;
 *   theta -= (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps),
 * moments and theta in float32, the products with the double hyper-parameters in double (as torch's fused kernel).
 * A NaN / +-Inf loss leaves theta, the moments and t untouched ("Skipping update", :147-148).
 *   theta dev [P] float32 (in/out); grad32 dev [P] float32 (out: the clipped gradient, what theta.grad holds);
 *   theta64 dev [P] float64 (out: the updated theta, the next epoch's circuit input);
 *   exp_avg, exp_avg_sq dev [P] float32 (in/out, zero before the first step);
 *   counters dev [2] int32 (in/out, zero before the first step): [0] = t, good steps so far; [1] = epochs so far;
 *   lr_table dev [n_lr] float64: the learning rate of epoch e at lr_table[min(e, n_lr - 1)] (the cosine schedule of
 *   :100-103 tabulated by the host; the epoch count advances on a skipped epoch too, like scheduler.step() at :158);
 *   loss_history dev [n_lr] float64, norm_history dev [n_lr] float32, or NULL: the same entry receives this epoch's
 *   loss and gradient norm (history['loss_ksd'] / ['grad_norm'], :163-166, without a copy per epoch). */
int bornvi_clip_adam_step(bornvi_handle h, int P, const double* grad64, double max_norm,
                          const double* loss, float* theta, float* grad32, double* theta64,
                          float* exp_avg, float* exp_avg_sq, int* counters, const double* lr_table,
                          int n_lr, double beta1, double beta2, double eps, float* total_norm,
                          double* loss_history, float* norm_history, bornvi_stream stream);

/* ---- adjoint differentiation: OPT-IN second gradient engine ---------------------------
 * The reference differentiates with diff_method="parameter-shift" (quantum_born_machine.py:58, :90, :114): 2P circuit
 * evaluations.  For L = f(q) the same gradient is  dL/dtheta_k = Im <lambda_k| P_k |phi_k>  (one forward and one
 * backward walk over the gates with two states, about three circuit evaluations).  Equal to the parameter-shift
 * gradient to rounding; never used unless the caller asks (it changes what "2P evaluations" means).
 *   bornvi_adjoint_state: theta dev [P] -> state dev [2^n] complex128 = U(theta)|0..0> (canonical order, wire 0 = MSB),
 *                         probs dev [2^n] = |state|^2 or NULL;
 *   bornvi_adjoint_vjp:   state (as returned above, left untouched), dLdq dev [2^n] -> grad dev [P]
 *                         = d/dtheta sum_z dLdq[z] q_z(theta).  Deterministic (fixed summation order). */
size_t bornvi_adjoint_workspace_bytes(bornvi_handle h, int ansatz, int n, int layers);
int bornvi_adjoint_state(bornvi_handle h, int ansatz, int n, int layers, const double* theta, double* state,
                         double* probs, void* workspace, size_t workspace_bytes, bornvi_stream stream);
int bornvi_adjoint_vjp(bornvi_handle h, int ansatz, int n, int layers, const double* theta, const double* state,
                       const double* dLdq, double* grad, void* workspace, size_t workspace_bytes,
                       bornvi_stream stream);

/* Diagnostic builds only (-DBORNVI_STAMPS=1, tools/probes): totals of the shader cycles wave 0 of every workgroup of the
 * fast circuit kernel spent per phase of its tile trips since the last call (out16[0..7]; out16[8] = workgroups counted);
 * synchronises the device.  All zeros in the production build. */
int bornvi_debug_circuit_stamps(bornvi_handle h, unsigned long long* out16);

/* ---- introspection (host only, no GPU needed): serialised execution plan of a circuit ------
 * (passes / stages / fused gates) as uint32 words; used by the CPU tests to check the
 * planner against the oracle.  Returns the number of words (writes min(cap, words)). */
long long bornvi_plan_describe(int ansatz, int n, int layers, int tile_bits, uint32_t* out,
                               size_t cap_words);

/* Host-only: first pass of the plan whose matrices depend on parameter p (what the opt-in prefix sharing orders a
 * parameter-shift batch by); returns the number of parameters, or -1 for an unsupported configuration. */
int bornvi_plan_param_first_pass(int ansatz, int n, int layers, int tile_bits, int* out, int cap);

/* The tables the fast pass kernel reads (one entry per stage, tile and thread: LDS slots with the CNOT
 * index maps folded in, CZ sign bits), derived on the host from the plan above; pass_off_out[i] = word
 * offset of pass i's header.  Returns the number of words, 0 when the plan runs on the generic kernel
 * (tiles below 2^10 amplitudes), -1 when unsupported.  Host only; used by the CPU tests. */
long long bornvi_plan_fast_describe(int ansatz, int n, int layers, int tile_bits, uint32_t* out,
                                    size_t cap_words, uint32_t* pass_off_out, int cap_passes);

/* `tile_bits` of the three functions above: bits 0-7 tile size (0 = default), bit 8 the planner's read_map option, bit 9
 * (0x200) the plan with 3 register wires per stage (8 amplitudes per thread) that the high-occupancy pass kernel runs.
 *
 * The COMPACT tables of that plan (every per-(tile row, thread) word of the fast tables is GF(2)-affine in (tile row,
 * thread): one word per lane plus one per (tile row, wave)); same return convention.  Host only; CPU tests. */
long long bornvi_plan_compact_describe(int ansatz, int n, int layers, int tile_bits, uint32_t* out,
                                       size_t cap_words, uint32_t* pass_off_out, int cap_passes);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* BORNVI_H */
