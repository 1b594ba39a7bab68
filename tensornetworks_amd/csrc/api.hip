// C ABI of libbornvi_hip.so (include/bornvi.h): argument checking, plan cache, workspace carving,
// kernel sequencing.  No exceptions leave this file; HIP errors are captured into the handle.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "bornvi.h"
#include "kernels.hpp"
#include "plan.hpp"

using namespace bornvi;

namespace {

struct DevPlan {
  Plan plan;
  uint32_t* d_words = nullptr;
  // fast path (plan.hpp: build_fast_tables): per-(stage, tile, thread) tables + persistent grid size
  FastTables fast;
  uint32_t* d_fast = nullptr;
  int fast_workgroups = 0;   // co-resident workgroups of circuit_pass_fast_kernel on the whole device
  // plans with 3 register wires (8 amplitudes per thread): compact tables + circuit_pass_r3_kernel
  CompactTables compact;
  uint32_t* d_compact = nullptr;
  size_t r3_lds = 0;
  int r3_workgroups = 0;
  // prefix sharing of parameter-shift batches (circuit_batch): device tables per (parameter range, chunk capacity)
  struct ShareChunk {
    int bc = 0;                        // circuits in the chunk, slot 0 = the base circuit
    std::vector<int> active, fresh;    // per pass: circuits that run / first circuit that reads the base state
    int* d_tab = nullptr;              // [bc] shift codes (launch_build_gates), then [bc] output rows
  };
  struct ShareTables {
    int p_begin = 0, p_count = 0, p_stride = 1, include_base = 0;
    long long bc_max = 0;
    std::vector<ShareChunk> chunks;
  };
  std::vector<std::unique_ptr<ShareTables>> share_cache;
};


// Block sequence of the adjoint walk (kernels_adjoint.hip) for one (ansatz, n, layers): rotation blocks and entangler
// blocks in program order; built on the host from the same gate list as the plans.
struct AdjPlan {
  struct Step { int is_rot; int index; };        // index into rots / ents
  std::vector<Step> steps;
  std::vector<AdjRotBlock> rots;
  std::vector<AdjEntangler> ents_fwd, ents_bwd;  // psi'[y] = s(y) psi[A y]  /  psi[x] = s(y) psi'[y], y = A^-1 x
  int* d_slot_param = nullptr;                   // [4 * rots.size()] parameter of (rotation block, element) or -1
  int n_params = 0;
};

thread_local std::string g_create_error;

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

struct bornvi_ctx {
  int device = 0;
  std::string err;
  PlanOptions opt;
  std::map<std::tuple<int, int, int>, std::unique_ptr<DevPlan>> plans;  // (ansatz | -1 = kron, n, layers)
  std::map<std::tuple<int, int, int>, std::unique_ptr<AdjPlan>> adj_plans;
  size_t max_lds_prepared = 0;
  size_t max_r3_lds_prepared = 0;
  int debug_flags = 0;  // timing-only ablations of circuit_pass_kernel (results are WRONG when non-zero)
  int num_cus = 256;    // multiProcessorCount
  int circuit_cus = 0;  // > 0: size the persistent circuit grid for this many CUs (the caller launches on a CU-masked stream)
  int wgs_per_cu = 0;   // generic kernel: > 0 = persistent grid of num_cus * wgs_per_cu workgroups; 0 = one per tile
  int fast_path = 1;    // 1: circuit_pass_fast_kernel where the plan is eligible; 0: always the generic kernel
  int fast_wgs_per_cu = 0;  // fast kernel: 0 = what the occupancy query admits
  int prefix_share = 0;     // OPT-IN: in a parameter-shift batch a shifted circuit starts from the base
                            // circuit's state before the first pass its parameter touches -- bit-identical rows, ~40 % fewer
                            // circuit-passes.  Off by default: the north-star path is 2P full circuit evaluations.
  int grad_engine = 0;      // OPT-IN: 1 = bornvi_paramshift_grad answers with adjoint differentiation (one
                            // forward + one backward walk) instead of 2P shifted circuits; same gradient to rounding
  int batched_quadform = 1; // bornvi_stein_quadform with B > 1: 1 = one MFMA pass over K, 0 = B GEMV passes (A/B switch)
  int zero_support = 1;     // fast kernel: pass 0 / pass 1 leave out what the support of |0..0> makes known zeros (A/B)
  int alternate_walk = 1;   // fast kernel: odd passes walk the tiles from the last to the first: a pass starts on the states the previous
                            // one wrote last, which the 256 MiB memory-side cache still holds (n = 16: -4.4 %; n = 20: neutral)
  int direct_stages = 3;    // fast kernel: bit 0 / 1 = first / last stage of a pass straight from / to HBM where the plan allows
                            // (A/B switch; bits 2.. = 1 + the only pass allowed to, for debugging)
};

namespace {

int fail(bornvi_handle h, int code, const std::string& msg) {
  if (h) h->err = msg;
  return code;
}

#define HIPCHK(h, call)                                                                              \
  do {                                                                                               \
    hipError_t e_ = (call);                                                                          \
    if (e_ != hipSuccess)                                                                            \
      return fail(h, BORNVI_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));             \
  } while (0)

// Entry points run on the handle's device and leave the calling thread's current device as they found it (a process
// may drive several GPUs through several handles, and torch tracks the current device itself).
struct DeviceScope {
  int prev = -1;
  bool changed = false;
  hipError_t err = hipSuccess;
  explicit DeviceScope(int dev) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) { err = hipSetDevice(dev); changed = (err == hipSuccess); }
  }
  ~DeviceScope() { if (changed) (void)hipSetDevice(prev); }
  DeviceScope(const DeviceScope&) = delete;
  DeviceScope& operator=(const DeviceScope&) = delete;
};
#define DEVICE_SCOPE(h)                    \
  DeviceScope dev_scope_((h)->device);     \
  HIPCHK(h, dev_scope_.err)

// Plans, tables and share tables are uploaded through the NULL stream; the kernels that read them run on the caller's
// stream, which (torch's streams) does not synchronise with it: one device-wide wait per NEW plan / table.  That wait is
// illegal while a stream capture is active -- a captured region must find its plans built (run the call once eagerly
// before capturing, as make_graphed_step and the adversarial epoch graph do); say so instead of a bare HIP error.
int sync_uploads(bornvi_handle h) {
  const hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) return BORNVI_OK;
  if (e == hipErrorStreamCaptureUnsupported || e == hipErrorStreamCaptureInvalidated || e == hipErrorStreamCaptureImplicit)
    return fail(h, BORNVI_ERR_HIP, std::string("a new circuit plan or table had to be built while a stream capture was active (") +
                                       hipGetErrorString(e) + "): run this call once eagerly with the same options before capturing");
  return fail(h, BORNVI_ERR_HIP, std::string("hipDeviceSynchronize: ") + hipGetErrorString(e));
}

int get_plan(bornvi_handle h, int ansatz, int n, int layers, DevPlan** out) {
  auto key = std::make_tuple(ansatz, n, layers);
  auto it = h->plans.find(key);
  if (it != h->plans.end()) { *out = it->second.get(); return BORNVI_OK; }
  auto dp = std::make_unique<DevPlan>();
  std::string msg;
  bool ok = false, use_r3 = false;
  if (h->opt.r == 3) {
    // 8 amplitudes per thread where the plan is eligible for circuit_pass_r3_kernel (compact tables, tile and tables
    // within the CU's LDS); otherwise the 16-amplitude plan below
    std::string m3;
    ok = (ansatz == -1) ? make_kron_plan(n, h->opt, dp->plan, m3) : make_plan(ansatz, n, layers, h->opt, dp->plan, m3);
    use_r3 = ok && dp->plan.r == 3 && build_compact_tables(dp->plan, dp->compact, m3) &&
             dp->compact.lds_bytes(dp->plan.k) <= MAX_LDS_BYTES;
    if (!use_r3) { dp = std::make_unique<DevPlan>(); ok = false; }
  }
  if (!ok) {
    PlanOptions o4 = h->opt;
    o4.r = 4;
    if (o4.max_threads > 512) o4.max_threads = 512;
    ok = (ansatz == -1) ? make_kron_plan(n, o4, dp->plan, msg) : make_plan(ansatz, n, layers, o4, dp->plan, msg);
  }
  if (!ok) return fail(h, BORNVI_ERR_UNSUPPORTED, msg);
  DEVICE_SCOPE(h);
  if (use_r3) {
    HIPCHK(h, hipMalloc((void**)&dp->d_words, dp->plan.words.size() * sizeof(uint32_t)));
    HIPCHK(h, hipMemcpy(dp->d_words, dp->plan.words.data(), dp->plan.words.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIPCHK(h, hipMalloc((void**)&dp->d_compact, dp->compact.words.size() * sizeof(uint32_t)));
    HIPCHK(h, hipMemcpy(dp->d_compact, dp->compact.words.data(), dp->compact.words.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    dp->r3_lds = dp->compact.lds_bytes(dp->plan.k);
    if (dp->r3_lds > h->max_r3_lds_prepared) {
      HIPCHK(h, prepare_circuit_r3_kernel(dp->r3_lds));
      h->max_r3_lds_prepared = dp->r3_lds;
    }
    dp->r3_workgroups = circuit_r3_workgroups_per_cu(dp->plan.threads, dp->r3_lds) * h->num_cus;
    if (dp->r3_workgroups <= 0) return fail(h, BORNVI_ERR_HIP, "circuit_pass_r3_kernel: no workgroup fits a CU");
    std::vector<uint32_t>().swap(dp->compact.words);
    { int rc_ = sync_uploads(h); if (rc_) return rc_; }
    *out = dp.get();
    h->plans[key] = std::move(dp);
    return BORNVI_OK;
  }
  HIPCHK(h, hipMalloc((void**)&dp->d_words, dp->plan.words.size() * sizeof(uint32_t)));
  HIPCHK(h, hipMemcpy(dp->d_words, dp->plan.words.data(), dp->plan.words.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  if (dp->plan.lds_bytes() > h->max_lds_prepared) {
    HIPCHK(h, prepare_circuit_kernel(dp->plan.lds_bytes()));
    h->max_lds_prepared = dp->plan.lds_bytes();
  }
  // fast path only when the tile, the matrices and one tile row of the stage tables fit the CU's 160 KiB of LDS
  if (build_fast_tables(dp->plan, FAST_TABLE_MAX_BYTES, dp->fast) &&
      dp->plan.fast_lds_bytes(dp->fast.max_tab_rows) <= MAX_LDS_BYTES) {
    HIPCHK(h, hipMalloc((void**)&dp->d_fast, dp->fast.words.size() * sizeof(uint32_t)));
    HIPCHK(h, hipMemcpy(dp->d_fast, dp->fast.words.data(), dp->fast.words.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    const size_t flds = dp->plan.fast_lds_bytes(dp->fast.max_tab_rows);
    if (flds > h->max_lds_prepared) {
      HIPCHK(h, prepare_circuit_kernel(flds));
      h->max_lds_prepared = flds;
    }
    dp->fast_workgroups = circuit_fast_workgroups_per_cu(1 << (dp->plan.k - 4), flds) * h->num_cus;
    std::vector<uint32_t>().swap(dp->fast.words);   // the host copy is no longer needed (pass_off is)
  }
  // The uploads above go through the NULL stream; the kernels that read them are launched on the caller's stream, which
  // (torch's streams) does not synchronise with it.  One device-wide wait per NEW plan.
  { int rc_ = sync_uploads(h); if (rc_) return rc_; }
  *out = dp.get();
  h->plans[key] = std::move(dp);
  return BORNVI_OK;
}

void free_plan(DevPlan* dp) {
  if (!dp) return;
  if (dp->d_words) (void)hipFree(dp->d_words);
  if (dp->d_fast) (void)hipFree(dp->d_fast);
  if (dp->d_compact) (void)hipFree(dp->d_compact);
  dp->d_words = dp->d_fast = dp->d_compact = nullptr;
  for (auto& stb : dp->share_cache)
    for (auto& c : stb->chunks) if (c.d_tab) (void)hipFree(c.d_tab);
  dp->share_cache.clear();
}

// bytes one circuit needs in the workspace: its fused-gate matrices + two ping-pong states
// slots of one circuit in the gate array: its fused matrices, then the scale of its probabilities (normalised gates)
inline int gate_slots(const Plan& p) { return p.n_fused + 1; }

size_t per_circuit_bytes(const Plan& p) {
  size_t b = (size_t)gate_slots(p) * 64;
  if (p.n_passes > 1) b += 2 * ((size_t)16 << p.n);
  return b ? b : 64;      // (a circuit without gates -- `basic` with 0 layers -- needs nothing; callers divide by this)
}

// Runs all passes of `dp` for `bc` circuits.  in0: input state of pass 0 (or null for |0..0>);
// bufA/bufB: ping-pong buffers; final_state / final_probs: destination of the last pass.
// share (fast path only): see DevPlan::ShareChunk -- pass i runs circuits [0, active[i]), those from fresh[i] on
// read slot 0 (the base circuit) of the input buffer; the last pass writes row d_tab[bc + b] of final_probs.
// pass_begin / pass_end: the passes to run (default: all); `in0` is then the input of pass `pass_begin` (the ping-pong
// continues from there: pass_input_buffer).  wdot / partials (8-amplitude kernel, last pass only): the fused dot product
// with dL/dq instead of the probabilities (kernels_circuit8.hip).
int run_passes(bornvi_handle h, DevPlan* dp, int bc, const void* in0, void* bufA, void* bufB, void* final_state,
               double* final_probs, const double* gates, long long gate_stride, hipStream_t st,
               const DevPlan::ShareChunk* share = nullptr, double* trash = nullptr, int pass_begin = 0, int pass_end = -1,
               const double* wdot = nullptr, double* partials = nullptr) {
  const Plan& p = dp->plan;
  const void* in = in0;
  if (pass_end < 0) pass_end = p.n_passes;
  for (int i = pass_begin; i < pass_end; ++i) {
    const bool last = (i == p.n_passes - 1);
    void* out = last ? final_state : ((in == bufA) ? bufB : bufA);
    if (dp->d_compact) {
      const int cus = h->circuit_cus > 0 ? h->circuit_cus : h->num_cus;
      const int wgs = h->fast_wgs_per_cu > 0 ? h->fast_wgs_per_cu * cus : dp->r3_workgroups / h->num_cus * cus;
      PrefixShare ps;
      int nb = bc;
      if (share) {
        nb = share->active[i];
        ps.fresh_begin = share->fresh[i];
        ps.row_map = share->d_tab + share->bc;
        ps.trash = trash;
      }
      HIPCHK(h, launch_circuit_pass_r3(dp->d_words, p.pass_off[i], dp->d_compact, dp->compact.pass_off[i], p.n, p.k, dp->r3_lds, nb, in,
                                       out, final_probs, gates, gate_stride, wgs,
                                       (((h->direct_stages >> 2) && (h->direct_stages >> 2) - 1 != i) ? 0 : (h->direct_stages & 3)) |
                                           ((h->alternate_walk && (i & 1)) ? 4 : 0) |
                                           ((pass_begin == 0 && in0 == nullptr && h->direct_stages == 3 && h->zero_support) ? 8 : 0),
                                       ps, last ? wdot : nullptr, last ? partials : nullptr, st));
    } else if (dp->d_fast && h->fast_path && dp->fast_workgroups > 0) {
      const int cus = h->circuit_cus > 0 ? h->circuit_cus : h->num_cus;
      const int wgs = h->fast_wgs_per_cu > 0 ? h->fast_wgs_per_cu * cus : dp->fast_workgroups / h->num_cus * cus;
      PrefixShare ps;
      int nb = bc;
      if (share) {
        nb = share->active[i];
        ps.fresh_begin = share->fresh[i];
        ps.row_map = share->d_tab + share->bc;
        ps.trash = trash;
      }
      HIPCHK(h, launch_circuit_pass_fast(dp->d_words, p.pass_off[i], dp->d_fast, dp->fast.pass_off[i], p.n, p.k,
                                         p.fast_lds_bytes(dp->fast.max_tab_rows), nb, in, out, final_probs, gates, gate_stride, wgs,
                                         p.fast_lds_tab_off(), p.fast_lds_mats2_off(dp->fast.max_tab_rows),
                                         (((h->direct_stages >> 2) && (h->direct_stages >> 2) - 1 != i) ? 0 : (h->direct_stages & 3)) |
                                             ((h->alternate_walk && (i & 1)) ? 4 : 0) |
                                             // (the zero-support masks of passes 0 and 1 assume both run with their direct
                                             // first stage on and without debug ablations: plan.cpp FH_ZINFO)
                                             ((in0 == nullptr && h->direct_stages == 3 && h->debug_flags == 0 && h->zero_support) ? 8 : 0),
                                         h->debug_flags, ps, st));
    } else {
      if (share) return fail(h, BORNVI_ERR_INVALID, "prefix sharing needs the fast circuit kernel");
      HIPCHK(h, launch_circuit_pass(dp->d_words, p.pass_off[i], p.n, p.k, p.threads, p.lds_bytes(), bc, in, out, final_probs, gates, gate_stride, h->wgs_per_cu * h->num_cus, h->debug_flags, st));
    }
    in = out;
  }
  return BORNVI_OK;
}

// Device tables of a parameter-shift batch with prefix sharing: the 2 (p_end - p_begin) shifted circuits ordered by
// the first pass their parameter touches (Plan::param_first_pass), cut into chunks of at most bc_max circuits, each
// led by its own copy of the base circuit.  Built once per (range, capacity) and kept with the plan.
int get_share_tables(bornvi_handle h, DevPlan* dp, int p_begin, int p_count, int p_stride, int include_base, long long bc_max,
                     const DevPlan::ShareTables** out) {
  for (auto& t : dp->share_cache)
    if (t->p_begin == p_begin && t->p_count == p_count && t->p_stride == p_stride && t->include_base == include_base &&
        t->bc_max == bc_max) { *out = t.get(); return BORNVI_OK; }
  const Plan& p = dp->plan;
  if (dp->share_cache.size() >= 16) {          // (hipFree waits for the device: nothing in flight reads these)
    for (auto& c : dp->share_cache.front()->chunks) if (c.d_tab) (void)hipFree(c.d_tab);
    dp->share_cache.erase(dp->share_cache.begin());
  }
  std::vector<int> codes;
  for (int i = 0; i < p_count; ++i) { const int q = p_begin + i * p_stride; codes.push_back(2 * q); codes.push_back(2 * q + 1); }
  std::stable_sort(codes.begin(), codes.end(), [&](int a, int b) { return p.param_first_pass[a >> 1] < p.param_first_pass[b >> 1]; });
  auto tabs = std::make_unique<DevPlan::ShareTables>();
  tabs->p_begin = p_begin; tabs->p_count = p_count; tabs->p_stride = p_stride; tabs->include_base = include_base; tabs->bc_max = bc_max;
  const long long per_chunk = bc_max - 1;
  for (size_t c0 = 0; c0 < codes.size() || tabs->chunks.empty(); c0 += (size_t)per_chunk) {
    DevPlan::ShareChunk ch;
    const size_t c1 = std::min(codes.size(), c0 + (size_t)per_chunk);
    ch.bc = 1 + (int)(c1 - c0);
    std::vector<int> host(2 * (size_t)ch.bc);
    host[0] = -1;
    host[ch.bc] = (include_base && c0 == 0) ? 0 : -1;
    for (size_t c = c0; c < c1; ++c) {
      host[1 + (c - c0)] = codes[c];
      host[ch.bc + 1 + (c - c0)] = include_base + 2 * (((codes[c] >> 1) - p_begin) / p_stride) + (codes[c] & 1);
    }
    ch.active.assign(p.n_passes, 1);
    ch.fresh.assign(p.n_passes, 1);
    for (int i = 0; i < p.n_passes; ++i)
      for (size_t c = c0; c < c1; ++c) {
        const int fp = p.param_first_pass[codes[c] >> 1];
        if (fp <= i) ++ch.active[i];
        if (fp < i) ++ch.fresh[i];
      }
    // the last pass must deliver every circuit's probabilities
    if (ch.active[p.n_passes - 1] != ch.bc) return fail(h, BORNVI_ERR_INVALID, "prefix sharing: a parameter is touched by no pass");
    HIPCHK(h, hipMalloc((void**)&ch.d_tab, host.size() * sizeof(int)));
    HIPCHK(h, hipMemcpy(ch.d_tab, host.data(), host.size() * sizeof(int), hipMemcpyHostToDevice));
    tabs->chunks.push_back(std::move(ch));
    if (codes.empty()) break;
  }
  { int rc_ = sync_uploads(h); if (rc_) return rc_; }      // (NULL-stream uploads, read by kernels on the caller's stream: see get_plan)
  *out = tabs.get();
  dp->share_cache.push_back(std::move(tabs));
  return BORNVI_OK;
}

int circuit_batch(bornvi_handle h, int ansatz, int n, int layers, long long batch, const double* thetas,
                  int shift_mode, int p_begin, int p_stride, int include_base, double* probs, void* ws, size_t ws_bytes,
                  hipStream_t st) {
  if (!h) return BORNVI_ERR_INVALID;
  if (batch < 0) return fail(h, BORNVI_ERR_INVALID, "negative batch");
  if (batch == 0) return BORNVI_OK;       // (an empty batch has no buffers: null pointers are fine)
  if ((!thetas && num_params(ansatz, n, layers) > 0) || !probs)
    return fail(h, BORNVI_ERR_INVALID, "null pointer");
  DevPlan* dp = nullptr;
  int rc = get_plan(h, ansatz, n, layers, &dp);
  if (rc) return rc;
  const Plan& p = dp->plan;
  const size_t pcb = per_circuit_bytes(p);
  if (!ws || ws_bytes < 512) return fail(h, BORNVI_ERR_WORKSPACE, "workspace missing");
  // carve: [gates | stateA | stateB], each region 256-byte aligned
  long long bc_max = (long long)((ws_bytes - 512) / pcb);
  if (bc_max > batch) bc_max = batch;
  if (bc_max > 65535) bc_max = 65535;  // gridDim.y
  if (bc_max < 1) return fail(h, BORNVI_ERR_WORKSPACE, "workspace too small for one circuit");
  char* base = (char*)ws;
  double* gates = (double*)base;
  const size_t gates_bytes = align_up((size_t)bc_max * gate_slots(p) * 64, 256);
  const int normalise = dp->d_compact ? (p.n_passes == 1 ? 2 : 1) : 0;   // (2: records without the scale kernel, see launch_build_gates)
  const size_t state_bytes = align_up((size_t)bc_max * ((size_t)16 << n), 256);
  void* bufA = base + gates_bytes;
  void* bufB = base + gates_bytes + state_bytes;
  DEVICE_SCOPE(h);
  if (shift_mode && h->prefix_share && p.n_passes > 1 && (dp->d_compact || (dp->d_fast && h->fast_path && dp->fast_workgroups > 0)) &&
      batch > include_base) {
    // parameter-shift batch with prefix sharing: [gates | stateA | stateB | trash row]; each chunk carries the
    // base circuit in slot 0, so one slot of the capacity goes to it
    const size_t trash_bytes = align_up((size_t)8 << n, 256);
    if (ws_bytes >= 1024 + trash_bytes + 2 * pcb) {
      long long cap = (long long)((ws_bytes - 1024 - trash_bytes) / pcb);
      const long long want = batch - include_base + 1;
      if (cap > want) cap = want;
      if (cap > 65535) cap = 65535;
      const DevPlan::ShareTables* tabs = nullptr;
      rc = get_share_tables(h, dp, p_begin, (int)((batch - include_base) / 2), p_stride, include_base, cap, &tabs);
      if (rc) return rc;
      const size_t gb = align_up((size_t)cap * gate_slots(p) * 64, 256);
      const size_t sb = align_up((size_t)cap * ((size_t)16 << n), 256);
      void* sA = base + gb;
      void* sB = base + gb + sb;
      double* trash = (double*)(base + gb + 2 * sb);
      for (const DevPlan::ShareChunk& ch : tabs->chunks) {
        HIPCHK(h, launch_build_gates(dp->d_words, p.n_fused, thetas, p.n_params, 1, 0, 1, 0, 0, ch.bc, gates, ch.d_tab, gate_slots(p), normalise, st));
        rc = run_passes(h, dp, ch.bc, nullptr, sA, sB, nullptr, probs, gates, (long long)gate_slots(p) * 8, st, &ch, trash);
        if (rc) return rc;
      }
      return BORNVI_OK;
    }
  }
  for (long long c0 = 0; c0 < batch; c0 += bc_max) {
    const int bc = (int)((batch - c0 < bc_max) ? batch - c0 : bc_max);
    HIPCHK(h, launch_build_gates(dp->d_words, p.n_fused, thetas, p.n_params, shift_mode, p_begin, p_stride, include_base, c0, bc, gates, nullptr, gate_slots(p), normalise, st));
    rc = run_passes(h, dp, bc, nullptr, bufA, bufB, nullptr, probs + (c0 << n), gates, (long long)gate_slots(p) * 8, st);
    if (rc) return rc;
  }
  return BORNVI_OK;
}

// ---- adjoint walk: gate list -> rotation / entangler blocks ------------------------------------------------------
bool build_adj_plan(int ansatz, int n, int layers, AdjPlan& out, std::string& msg) {
  std::vector<Gate> gates;
  if (!build_gate_list(ansatz, n, layers, gates)) { msg = "unknown ansatz"; return false; }
  if (n < 1 || n > 30) { msg = "adjoint engine supports 1 <= n <= 30"; return false; }
  out.n_params = num_params(ansatz, n, layers);
  std::vector<std::vector<Gate>> per_wire((size_t)n);     // one-qubit gates of the current segment, per wire (they commute across wires)
  std::vector<Gate> two;                                   // two-qubit gates of the current entangler
  auto flush_rots = [&]() {
    for (int w = 0; w < n; ++w) {
      auto& g = per_wire[(size_t)w];
      for (size_t i = 0; i < g.size(); i += 4) {
        AdjRotBlock b{};
        b.wire = w;
        b.nrot = (int)std::min<size_t>(4, g.size() - i);
        for (int e = 0; e < 4; ++e) { b.kind[e] = e < b.nrot ? g[i + e].kind : G_H; b.param[e] = e < b.nrot ? g[i + e].param : -1; }
        out.steps.push_back({1, (int)out.rots.size()});
        out.rots.push_back(b);
      }
      g.clear();
    }
  };
  auto flush_ent = [&]() -> bool {
    if (two.empty()) return true;
    AdjEntangler F{}, Bk{};
    unsigned L[32], Inv[32];
    for (int b = 0; b < 32; ++b) { L[b] = b < n ? 1u << b : 0u; Inv[b] = L[b]; }
    int ncz = 0;
    for (int k = (int)two.size() - 1; k >= 0; --k) {       // L = C_{k+1} ... C_m while gate k is looked at
      const int a = n - 1 - two[(size_t)k].w0, b = n - 1 - two[(size_t)k].w1;   // physical bits (wire 0 = MSB)
      if (two[(size_t)k].kind == G_CZ) {
        if (ncz >= ADJ_MAX_CZ) return false;
        F.za[ncz] = L[a]; F.zb[ncz] = L[b]; ++ncz;
      } else {
        L[b] ^= L[a];                                       // CNOT(control a, target b): x_b = y_b ^ y_a
      }
    }
    for (size_t k = 0; k < two.size(); ++k)
      if (two[k].kind == G_CNOT) { const int a = n - 1 - two[k].w0, b = n - 1 - two[k].w1; Inv[b] ^= Inv[a]; }
    for (int b = 0; b < 32; ++b) { F.row[b] = L[b]; Bk.row[b] = Inv[b]; }
    F.ncz = ncz; Bk.ncz = ncz;
    for (int k = 0; k < ncz; ++k) { Bk.za[k] = F.za[k]; Bk.zb[k] = F.zb[k]; }
    out.steps.push_back({0, (int)out.ents_fwd.size()});
    out.ents_fwd.push_back(F);
    out.ents_bwd.push_back(Bk);
    two.clear();
    return true;
  };
  for (const Gate& g : gates) {
    if (g.w1 < 0) {
      if (!two.empty() && !flush_ent()) { msg = "too many CZ gates in one entangler block"; return false; }
      per_wire[(size_t)g.w0].push_back(g);
    } else {
      flush_rots();
      two.push_back(g);
    }
  }
  flush_rots();
  if (!flush_ent()) { msg = "too many CZ gates in one entangler block"; return false; }
  // adj_reduce_kernel writes grad[p] once per slot: a parameter shared by two gates would be overwritten, not summed
  std::vector<int> uses((size_t)(out.n_params > 0 ? out.n_params : 0), 0);
  for (const AdjRotBlock& b : out.rots)
    for (int e = 0; e < b.nrot; ++e)
      if (b.param[e] >= 0) {
        if (b.param[e] >= out.n_params || ++uses[(size_t)b.param[e]] > 1) { msg = "adjoint engine: a parameter is carried by more than one gate"; return false; }
      }
  return true;
}

int get_adj_plan(bornvi_handle h, int ansatz, int n, int layers, AdjPlan** out) {
  auto key = std::make_tuple(ansatz, n, layers);
  auto it = h->adj_plans.find(key);
  if (it != h->adj_plans.end()) { *out = it->second.get(); return BORNVI_OK; }
  auto ap = std::make_unique<AdjPlan>();
  std::string msg;
  if (!build_adj_plan(ansatz, n, layers, *ap, msg)) return fail(h, BORNVI_ERR_UNSUPPORTED, msg);
  std::vector<int> slots(4 * ap->rots.size() + 4, -1);
  for (size_t b = 0; b < ap->rots.size(); ++b)
    for (int e = 0; e < ap->rots[b].nrot; ++e) slots[4 * b + e] = ap->rots[b].param[e];
  DEVICE_SCOPE(h);
  HIPCHK(h, hipMalloc((void**)&ap->d_slot_param, slots.size() * sizeof(int)));
  HIPCHK(h, hipMemcpy(ap->d_slot_param, slots.data(), slots.size() * sizeof(int), hipMemcpyHostToDevice));
  { int rc_ = sync_uploads(h); if (rc_) return rc_; }      // (NULL-stream uploads, read by kernels on the caller's stream: see get_plan)
  *out = ap.get();
  h->adj_plans[key] = std::move(ap);
  return BORNVI_OK;
}

bool valid_n_for_dense(int n) { return n >= 1 && n <= 17; }
bool valid_ld(int n, long long ld) { return ld >= (1ll << n) && (ld & 1) == 0 && ld <= (1ll << n) + 4096; }

}  // namespace

extern "C" {

int bornvi_version(void) { return BORNVI_VERSION; }

int bornvi_create(int device_ordinal, bornvi_handle* out) {
  if (!out) return BORNVI_ERR_INVALID;
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    g_create_error = std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    return BORNVI_ERR_HIP;
  }
  if (device_ordinal < 0 || device_ordinal >= count) {
    g_create_error = "device ordinal out of range";
    return BORNVI_ERR_INVALID;
  }
  bornvi_ctx* h = new (std::nothrow) bornvi_ctx();
  if (!h) { g_create_error = "out of host memory"; return BORNVI_ERR_INVALID; }
  h->device = device_ordinal;
  // (A/B switches for whole test / bench runs; bornvi_set_option "reg_wires" / "read_map" are the per-handle form)
  if (const char* e = std::getenv("BORNVI_REG_WIRES")) { if (e[0] == '3' || e[0] == '4') h->opt.r = e[0] - '0'; }
  if (const char* e = std::getenv("BORNVI_CONTIG_OUT")) h->opt.contig_out = e[0] != '0';
  if (const char* e = std::getenv("BORNVI_READ_MAP")) h->opt.read_map = e[0] == '1' ? 1 : (e[0] == '0' ? 0 : -1);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess && prop.multiProcessorCount > 0)
    h->num_cus = prop.multiProcessorCount;
  *out = h;
  return BORNVI_OK;
}

void bornvi_destroy(bornvi_handle h) {
  if (!h) return;
  for (auto& kv : h->plans) free_plan(kv.second.get());
  for (auto& kv : h->adj_plans) if (kv.second->d_slot_param) (void)hipFree(kv.second->d_slot_param);
  delete h;
}

const char* bornvi_last_error(bornvi_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int bornvi_set_option(bornvi_handle h, const char* name, long long value) {
  if (!h || !name) return BORNVI_ERR_INVALID;
  if (!std::strcmp(name, "debug_flags")) { h->debug_flags = (int)value; return BORNVI_OK; }
  if (!std::strcmp(name, "workgroups_per_cu")) {
    if (value < 0 || value > 16) return fail(h, BORNVI_ERR_INVALID, "option value out of range");
    h->wgs_per_cu = (int)value;
    return BORNVI_OK;
  }
  if (!std::strcmp(name, "fast_path")) { h->fast_path = value ? 1 : 0; return BORNVI_OK; }
  if (!std::strcmp(name, "circuit_cus")) { h->circuit_cus = (int)value; return BORNVI_OK; }
  if (!std::strcmp(name, "prefix_share")) { h->prefix_share = (int)value; return BORNVI_OK; }
  if (!std::strcmp(name, "batched_quadform")) { h->batched_quadform = value ? 1 : 0; return BORNVI_OK; }
  if (!std::strcmp(name, "grad_engine")) {
    if (value != 0 && value != 1) return fail(h, BORNVI_ERR_INVALID, "grad_engine: 0 = parameter shift, 1 = adjoint");
    h->grad_engine = (int)value;
    return BORNVI_OK;
  }
  if (!std::strcmp(name, "zero_support")) { h->zero_support = value != 0; return BORNVI_OK; }
  if (!std::strcmp(name, "alternate_walk")) { h->alternate_walk = value != 0; return BORNVI_OK; }
  if (!std::strcmp(name, "direct_stages")) { h->direct_stages = (int)value; return BORNVI_OK; }   // bits 2..: 1 + the only pass allowed (debug)
  if (!std::strcmp(name, "fast_workgroups_per_cu")) {
    if (value < 0 || value > 16) return fail(h, BORNVI_ERR_INVALID, "option value out of range");
    h->fast_wgs_per_cu = (int)value;
    return BORNVI_OK;
  }
  PlanOptions o = h->opt;
  if (!std::strcmp(name, "tile_bits")) { o.kmax = (int)value; o.kmulti = (int)value; }
  else if (!std::strcmp(name, "tile_bits_multi")) o.kmulti = (int)value;
  else if (!std::strcmp(name, "low_bits")) o.lo = (int)value;
  else if (!std::strcmp(name, "max_threads")) o.max_threads = (int)value;
  else if (!std::strcmp(name, "read_map")) o.read_map = value < 0 ? -1 : (value != 0 ? 1 : 0);   // -1: by the kernel (on with reg_wires = 3)
  else if (!std::strcmp(name, "reg_wires")) o.r = (int)value;
  else if (!std::strcmp(name, "contig_out")) o.contig_out = value != 0;
  else return fail(h, BORNVI_ERR_INVALID, std::string("unknown option ") + name);
  if (o.kmax < 4 || o.kmax > 13 || (o.kmulti != 0 && (o.kmulti < 4 || o.kmulti > 13)) || o.lo < 0 || o.lo > 8 || o.max_threads < 64 || o.max_threads > 1024 ||
      (o.max_threads & (o.max_threads - 1)) || (o.r != 3 && o.r != 4))
    return fail(h, BORNVI_ERR_INVALID, "option value out of range");
  h->opt = o;
  for (auto& kv : h->plans) free_plan(kv.second.get());
  h->plans.clear();
  return BORNVI_OK;
}

int bornvi_get_option(bornvi_handle h, const char* name, long long* value) {
  if (!h || !name || !value) return BORNVI_ERR_INVALID;
  if (!std::strcmp(name, "reg_wires")) *value = h->opt.r;
  else if (!std::strcmp(name, "read_map")) *value = h->opt.read_map;
  else if (!std::strcmp(name, "tile_bits")) *value = h->opt.kmax;
  else if (!std::strcmp(name, "tile_bits_multi")) *value = h->opt.kmulti;
  else if (!std::strcmp(name, "low_bits")) *value = h->opt.lo;
  else if (!std::strcmp(name, "contig_out")) *value = h->opt.contig_out;
  else if (!std::strcmp(name, "prefix_share")) *value = h->prefix_share;
  else if (!std::strcmp(name, "grad_engine")) *value = h->grad_engine;
  else if (!std::strcmp(name, "fast_path")) *value = h->fast_path;
  else if (!std::strcmp(name, "direct_stages")) *value = h->direct_stages;
  else if (!std::strcmp(name, "zero_support")) *value = h->zero_support;
  else if (!std::strcmp(name, "alternate_walk")) *value = h->alternate_walk;
  else if (!std::strcmp(name, "batched_quadform")) *value = h->batched_quadform;
  else return fail(h, BORNVI_ERR_INVALID, std::string("unknown option ") + name);
  return BORNVI_OK;
}

int bornvi_num_params(int ansatz, int n, int layers) { return num_params(ansatz, n, layers); }

int bornvi_num_gates(int ansatz, int n, int layers) {
  std::vector<Gate> g;
  if (!build_gate_list(ansatz, n, layers, g)) return -1;
  return (int)g.size();
}

size_t bornvi_circuit_workspace_bytes(bornvi_handle h, int ansatz, int n, int layers, int batch) {
  if (!h || batch < 0) return 0;
  DevPlan* dp = nullptr;
  if (get_plan(h, ansatz, n, layers, &dp)) return 0;
  const Plan& p = dp->plan;
  // (one circuit and one probability row more than `batch`: a parameter-shift batch with prefix sharing keeps the
  // base circuit in slot 0 even when the caller does not ask for its row, and a row for unwanted output)
  const size_t bb = (size_t)batch + (p.n_passes > 1 ? 1 : 0);
  size_t b = 1024 + align_up(bb * gate_slots(p) * 64, 256);
  if (p.n_passes > 1) b += 2 * align_up(bb * ((size_t)16 << n), 256) + align_up((size_t)8 << n, 256);
  return b;
}

int bornvi_stream_create_cu_range(bornvi_handle h, int first_cu, int num_cus, bornvi_stream* out) {
  if (!h || !out) return BORNVI_ERR_INVALID;
  *out = nullptr;
  if (first_cu < 0 || num_cus < 1 || first_cu + num_cus > h->num_cus) return fail(h, BORNVI_ERR_INVALID, "CU range out of bounds");
  DEVICE_SCOPE(h);
  std::vector<uint32_t> mask((size_t)(h->num_cus + 31) / 32, 0u);
  for (int c = first_cu; c < first_cu + num_cus; ++c) mask[(size_t)c / 32] |= 1u << (c % 32);
  hipStream_t st = nullptr;
  HIPCHK(h, hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
  *out = (bornvi_stream)st;
  return BORNVI_OK;
}

int bornvi_stream_destroy(bornvi_handle h, bornvi_stream stream) {
  if (!h) return BORNVI_ERR_INVALID;
  if (!stream) return BORNVI_OK;
  DEVICE_SCOPE(h);
  HIPCHK(h, hipStreamDestroy((hipStream_t)stream));
  return BORNVI_OK;
}

int bornvi_circuit_probs(bornvi_handle h, int ansatz, int n, int layers, int batch, const double* thetas,
                         double* probs, void* workspace, size_t workspace_bytes, bornvi_stream stream) {
  return circuit_batch(h, ansatz, n, layers, batch, thetas, 0, 0, 1, 0, probs, workspace, workspace_bytes, (hipStream_t)stream);
}

int bornvi_paramshift_probs_strided(bornvi_handle h, int ansatz, int n, int layers, const double* theta, int p_begin,
                                    int p_count, int p_stride, int include_base, double* probs, void* workspace,
                                    size_t workspace_bytes, bornvi_stream stream) {
  if (!h) return BORNVI_ERR_INVALID;
  const int P = num_params(ansatz, n, layers);
  if (P < 0 || p_begin < 0 || p_count < 0 || p_stride < 1 || (p_count > 0 && (long long)p_begin + (long long)(p_count - 1) * p_stride >= P))
    return fail(h, BORNVI_ERR_INVALID, "parameter range out of bounds");
  const long long batch = (include_base ? 1 : 0) + 2ll * p_count;
  return circuit_batch(h, ansatz, n, layers, batch, theta, 1, p_begin, p_stride, include_base ? 1 : 0, probs, workspace,
                       workspace_bytes, (hipStream_t)stream);
}

int bornvi_paramshift_probs(bornvi_handle h, int ansatz, int n, int layers, const double* theta, int p_begin,
                            int p_end, int include_base, double* probs, void* workspace, size_t workspace_bytes,
                            bornvi_stream stream) {
  if (!h) return BORNVI_ERR_INVALID;
  if (p_end < p_begin) return fail(h, BORNVI_ERR_INVALID, "parameter range out of bounds");
  return bornvi_paramshift_probs_strided(h, ansatz, n, layers, theta, p_begin, p_end - p_begin, 1, include_base, probs,
                                         workspace, workspace_bytes, stream);
}

size_t bornvi_paramshift_grad_workspace_bytes(bornvi_handle h, int ansatz, int n, int layers, int p_begin, int p_end) {
  if (!h || p_end < p_begin) return 0;
  const int nb = 2 * (p_end - p_begin);
  const size_t c = bornvi_circuit_workspace_bytes(h, ansatz, n, layers, nb);
  if (!c) return 0;
  size_t b = c + align_up((size_t)nb * ((size_t)8 << n), 256);
  if (h->grad_engine == 1) {      // adjoint: state + full gradient + the walk's workspace
    const int P = num_params(ansatz, n, layers);
    const size_t a = bornvi_adjoint_workspace_bytes(h, ansatz, n, layers);
    if (a) b = std::max(b, align_up((size_t)16 << n, 256) + align_up((size_t)(P > 0 ? P : 1) * 8, 256) + a);
  }
  return b;
}

int bornvi_paramshift_grad(bornvi_handle h, int ansatz, int n, int layers, const double* theta, const double* dLdq,
                           int p_begin, int p_end, double* grad, void* workspace, size_t workspace_bytes,
                           bornvi_stream stream) {
  if (!h) return BORNVI_ERR_INVALID;
  if (!theta || !dLdq || !grad) return fail(h, BORNVI_ERR_INVALID, "null pointer");
  const int ns = p_end - p_begin;
  if (ns == 0) return BORNVI_OK;
  if (h->grad_engine == 1) {
    const int P = num_params(ansatz, n, layers);
    if (P < 0 || p_begin < 0 || p_end > P) return fail(h, BORNVI_ERR_INVALID, "parameter range out of bounds");
    if (!workspace || workspace_bytes < bornvi_paramshift_grad_workspace_bytes(h, ansatz, n, layers, p_begin, p_end))
      return fail(h, BORNVI_ERR_WORKSPACE, "workspace too small");
    const size_t sb = align_up((size_t)16 << n, 256), gb = align_up((size_t)P * 8, 256);
    double* state = (double*)workspace;
    double* gfull = (double*)((char*)workspace + sb);
    void* aws = (char*)workspace + sb + gb;
    int rc = bornvi_adjoint_state(h, ansatz, n, layers, theta, state, nullptr, aws, workspace_bytes - sb - gb, stream);
    if (rc) return rc;
    rc = bornvi_adjoint_vjp(h, ansatz, n, layers, theta, state, dLdq, gfull, aws, workspace_bytes - sb - gb, stream);
    if (rc) return rc;
    DEVICE_SCOPE(h);
    HIPCHK(h, hipMemcpyAsync(grad, gfull + p_begin, (size_t)ns * 8, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return BORNVI_OK;
  }
  const size_t probs_bytes = align_up((size_t)2 * ns * ((size_t)8 << n), 256);
  if (!workspace || workspace_bytes < probs_bytes + 512) return fail(h, BORNVI_ERR_WORKSPACE, "workspace too small");
  double* shifted = (double*)workspace;
  int rc = bornvi_paramshift_probs(h, ansatz, n, layers, theta, p_begin, p_end, 0, shifted,
                                   (char*)workspace + probs_bytes, workspace_bytes - probs_bytes, stream);
  if (rc) return rc;
  HIPCHK(h, launch_shift_dot(shifted, ns, dLdq, nullptr, n, grad, nullptr, (hipStream_t)stream));
  return BORNVI_OK;
}

// ---- parameter-shift gradient with the dot product fused into the last circuit pass ------------------------------
namespace {
struct DotLayout { size_t gates_bytes, state_bytes, partial_bytes, total; long long bc; };
// workspace of a fused step: [gates | stateA | stateB | partials], all for 1 + 2 p_count circuits at once (no chunks)
bool dot_layout(const Plan& p, int n, int p_count, DotLayout& L) {
  L.bc = 1 + 2ll * p_count;
  if (L.bc > 65535) return false;
  L.gates_bytes = align_up((size_t)L.bc * gate_slots(p) * 64, 256);
  L.state_bytes = align_up((size_t)L.bc * ((size_t)16 << n), 256);
  L.partial_bytes = align_up((size_t)2 * p_count * ((size_t)8 << (n - p.k)), 256);
  L.total = 512 + L.gates_bytes + 2 * L.state_bytes + L.partial_bytes;
  return true;
}
// the fused path exists for multi-pass plans of the 8-amplitude kernel, without prefix sharing
bool dot_supported(bornvi_handle h, DevPlan* dp) {
  return dp->d_compact && dp->plan.n_passes >= 2 && dp->plan.k >= 9 && !h->prefix_share && h->grad_engine == 0;
}
// input buffer of pass i when pass 0 starts from |0..0> and writes bufA first
void* pass_input_buffer(int i, void* bufA, void* bufB) { return i == 0 ? nullptr : ((i & 1) ? bufA : bufB); }
}  // namespace

size_t bornvi_paramshift_dot_workspace_bytes(bornvi_handle h, int ansatz, int n, int layers, int p_count) {
  if (!h || p_count < 0) return 0;
  DevPlan* dp = nullptr;
  if (get_plan(h, ansatz, n, layers, &dp)) return 0;
  DotLayout L;
  if (!dot_supported(h, dp) || !dot_layout(dp->plan, n, p_count, L)) return 0;
  return L.total;
}

int bornvi_paramshift_dot_begin(bornvi_handle h, int ansatz, int n, int layers, const double* theta, int p_begin, int p_count,
                                int p_stride, double* q_out, void* workspace, size_t workspace_bytes, bornvi_stream stream) {
  if (!h) return BORNVI_ERR_INVALID;
  if (!theta || !q_out) return fail(h, BORNVI_ERR_INVALID, "null pointer");
  const int P = num_params(ansatz, n, layers);
  if (P < 0 || p_begin < 0 || p_count < 0 || p_stride < 1 || (p_count > 0 && (long long)p_begin + (long long)(p_count - 1) * p_stride >= P))
    return fail(h, BORNVI_ERR_INVALID, "parameter range out of bounds");
  DevPlan* dp = nullptr;
  int rc = get_plan(h, ansatz, n, layers, &dp);
  if (rc) return rc;
  DotLayout L;
  if (!dot_supported(h, dp) || !dot_layout(dp->plan, n, p_count, L))
    return fail(h, BORNVI_ERR_UNSUPPORTED, "the fused parameter-shift dot needs a multi-pass plan of the 8-amplitude kernel (reg_wires = 3), no prefix sharing");
  if (!workspace || workspace_bytes < L.total || ((uintptr_t)workspace & 15)) return fail(h, BORNVI_ERR_WORKSPACE, "workspace too small or misaligned");
  const Plan& p = dp->plan;
  hipStream_t st = (hipStream_t)stream;
  char* base = (char*)workspace;
  double* gates = (double*)base;
  void* bufA = base + L.gates_bytes;
  void* bufB = base + L.gates_bytes + L.state_bytes;
  DEVICE_SCOPE(h);
  // circuit 0 = the base circuit, then (+p, -p) for the p_count parameters p_begin, p_begin + p_stride, ...
  HIPCHK(h, launch_build_gates(dp->d_words, p.n_fused, theta, p.n_params, 1, p_begin, p_stride, 1, 0, (int)L.bc, gates, nullptr, gate_slots(p), 1, st));
  rc = run_passes(h, dp, (int)L.bc, nullptr, bufA, bufB, nullptr, nullptr, gates, (long long)gate_slots(p) * 8, st, nullptr, nullptr, 0, p.n_passes - 1);
  if (rc) return rc;
  // the base circuit's last pass alone: q
  void* in_last = pass_input_buffer(p.n_passes - 1, bufA, bufB);
  return run_passes(h, dp, 1, in_last, bufA, bufB, nullptr, q_out, gates, (long long)gate_slots(p) * 8, st, nullptr, nullptr, p.n_passes - 1, p.n_passes);
}

int bornvi_paramshift_dot_finish(bornvi_handle h, int ansatz, int n, int layers, int p_count, const double* w, const double* ksd2,
                                 double* grad, double* loss_out, void* workspace, size_t workspace_bytes, bornvi_stream stream) {
  if (!h) return BORNVI_ERR_INVALID;
  if (!w || p_count < 0 || (p_count > 0 && !grad)) return fail(h, BORNVI_ERR_INVALID, "bad argument");
  DevPlan* dp = nullptr;
  int rc = get_plan(h, ansatz, n, layers, &dp);
  if (rc) return rc;
  DotLayout L;
  if (!dot_supported(h, dp) || !dot_layout(dp->plan, n, p_count, L))
    return fail(h, BORNVI_ERR_UNSUPPORTED, "the fused parameter-shift dot needs a multi-pass plan of the 8-amplitude kernel (reg_wires = 3), no prefix sharing");
  if (!workspace || workspace_bytes < L.total || ((uintptr_t)workspace & 15)) return fail(h, BORNVI_ERR_WORKSPACE, "workspace too small or misaligned");
  const Plan& p = dp->plan;
  hipStream_t st = (hipStream_t)stream;
  char* base = (char*)workspace;
  double* gates = (double*)base;
  char* bufA = base + L.gates_bytes;
  char* bufB = base + L.gates_bytes + L.state_bytes;
  double* partials = (double*)(base + L.gates_bytes + 2 * L.state_bytes);
  DEVICE_SCOPE(h);
  if (p_count > 0) {
    // the shifted circuits' last pass (circuits 1 .. 2 p_count of the batch bornvi_paramshift_dot_begin left in the
    // workspace): no probabilities, per-(circuit, tile) shares of  sum_z w_z q(z)
    const size_t one_state = (size_t)16 << n;
    char* in_last = (char*)pass_input_buffer(p.n_passes - 1, bufA, bufB) + one_state;
    rc = run_passes(h, dp, 2 * p_count, in_last, bufA + one_state, bufB + one_state, nullptr, nullptr, gates + (size_t)gate_slots(p) * 8,
                    (long long)gate_slots(p) * 8, st, nullptr, nullptr, p.n_passes - 1, p.n_passes, w, partials);
    if (rc) return rc;
  }
  HIPCHK(h, launch_dot_finish(partials, p_count, 1ll << (n - p.k), ksd2, grad, loss_out, st));
  return BORNVI_OK;
}

int bornvi_gate1q_apply(bornvi_handle h, int n, long long batch, double* state, int wire, const double* U,
                        bornvi_stream stream) {
  if (!h) return BORNVI_ERR_INVALID;
  if (!state || !U || n < 1 || n > 40 || wire < 0 || wire >= n || batch < 0) return fail(h, BORNVI_ERR_INVALID, "bad argument");
  if (batch == 0) return BORNVI_OK;
  DEVICE_SCOPE(h);
  HIPCHK(h, launch_gate1q(state, n, batch, wire, U, (hipStream_t)stream));
  return BORNVI_OK;
}

int bornvi_cnot_apply(bornvi_handle h, int n, long long batch, double* state, int control, int target,
                      bornvi_stream stream) {
  if (!h) return BORNVI_ERR_INVALID;
  if (!state || n < 2 || n > 40 || control < 0 || control >= n || target < 0 || target >= n || control == target || batch < 0)
    return fail(h, BORNVI_ERR_INVALID, "bad argument");
  if (batch == 0) return BORNVI_OK;
  DEVICE_SCOPE(h);
  HIPCHK(h, launch_cnot(state, n, batch, control, target, (hipStream_t)stream));
  return BORNVI_OK;
}

int bornvi_born_probs(bornvi_handle h, int n, long long batch, const double* state, double* probs, bornvi_stream stream) {
  if (!h) return BORNVI_ERR_INVALID;
  if (!state || !probs || n < 0 || n > 40 || batch < 0) return fail(h, BORNVI_ERR_INVALID, "bad argument");
  if (batch == 0) return BORNVI_OK;
  DEVICE_SCOPE(h);
  HIPCHK(h, launch_born_probs(state, probs, n, batch, (hipStream_t)stream));
  return BORNVI_OK;
}

int bornvi_score_from_cpts(bornvi_handle h, const bornvi_bn_desc* bn, int n, double* S, double* pxz, bornvi_stream stream) {
  if (!h) return BORNVI_ERR_INVALID;
  if (!bn || !S || n < 1 || n > 30) return fail(h, BORNVI_ERR_INVALID, "bad argument");
  if (bn->num_nodes < 1 || bn->num_nodes > 64 || bn->max_parents < 1 || !bn->role || !bn->n_parents || !bn->parents ||
      !bn->cpt_off || !bn->cpt)
    return fail(h, BORNVI_ERR_INVALID, "bad network descriptor");
  DEVICE_SCOPE(h);
  HIPCHK(h, launch_score(*bn, n, S, pxz, (hipStream_t)stream));
  return BORNVI_OK;
}

long long bornvi_stein_gram_ld(int n) {
  if (n < 1 || n > 17) return 0;
  // 256 bytes of padding per row.  Same-box A/B of the band kernel at n = 16 (tools/probes/sym_probe.py, three
  // matrices held at once, twice): pitch 2^n: 2.62 / 2.83 / 2.63 and 2.83 / 2.62 / 2.63 ms by allocation; pitch
  // 2^n + 32: 2.57 / 2.57 / 2.56 and 2.57 / 2.57 / 2.56 ms -- faster, and the dependence on where the matrix landed
  // is gone.  (With a power-of-two pitch the same column of every row maps to the same HBM channel and bank, and a wave
  // streams 32 rows at the same column.)  Matrices below 2 GiB stay dense: the single-GPU contraction runs the
  // full-matrix kernel on them (kernels_stein.hip: SYM_FULL_BELOW_N).
  return n >= 14 ? (1ll << n) + 32 : (1ll << n);
}

int bornvi_stein_gram_build_rows_ld(bornvi_handle h, int n, double length_scale, const double* S, long long row_begin,
                                    long long row_end, double* K_rows, long long ld, bornvi_stream stream) {
  if (!h) return BORNVI_ERR_INVALID;
  if (!S || (!K_rows && row_end > row_begin)) return fail(h, BORNVI_ERR_INVALID, "null pointer");
  if (!valid_n_for_dense(n)) return fail(h, BORNVI_ERR_UNSUPPORTED, "dense Gram supports 1 <= n <= 17 (8 * 4^n bytes)");
  if (!(length_scale > 0.0)) return fail(h, BORNVI_ERR_INVALID, "length_scale must be positive");
  if (row_begin < 0 || row_end < row_begin || row_end > (1ll << n)) return fail(h, BORNVI_ERR_INVALID, "row range out of bounds");
  if (!valid_ld(n, ld)) return fail(h, BORNVI_ERR_INVALID, "leading dimension must be even, >= 2^n and <= 2^n + 4096");
  DEVICE_SCOPE(h);
  HIPCHK(h, launch_gram_build(n, length_scale, S, K_rows, row_begin, row_end, ld, (hipStream_t)stream));
  return BORNVI_OK;
}

int bornvi_stein_gram_build_rows(bornvi_handle h, int n, double length_scale, const double* S, long long row_begin,
                                 long long row_end, double* K_rows, bornvi_stream stream) {
  if (n < 0 || n > 40) return h ? fail(h, BORNVI_ERR_UNSUPPORTED, "dense Gram supports 1 <= n <= 17 (8 * 4^n bytes)") : BORNVI_ERR_INVALID;
  return bornvi_stein_gram_build_rows_ld(h, n, length_scale, S, row_begin, row_end, K_rows, 1ll << n, stream);
}

int bornvi_stein_gram_build(bornvi_handle h, int n, double length_scale, const double* S, double* K, bornvi_stream stream) {
  if (n < 0 || n > 40) return h ? fail(h, BORNVI_ERR_UNSUPPORTED, "dense Gram supports 1 <= n <= 17 (8 * 4^n bytes)") : BORNVI_ERR_INVALID;
  return bornvi_stein_gram_build_rows(h, n, length_scale, S, 0, 1ll << n, K, stream);
}

int bornvi_stein_kp_pairs(bornvi_handle h, int n, double length_scale, long long M, const long long* zi,
                          const long long* zj, const double* si, const double* sj, double* out, bornvi_stream stream) {
  if (!h) return BORNVI_ERR_INVALID;
  if (n < 1 || n > 30 || M < 0 || (M > 0 && (!zi || !zj || !si || !sj || !out))) return fail(h, BORNVI_ERR_INVALID, "bad argument");
  if (!(length_scale > 0.0)) return fail(h, BORNVI_ERR_INVALID, "length_scale must be positive");
  DEVICE_SCOPE(h);
  HIPCHK(h, launch_kp_pairs(n, length_scale, M, zi, zj, si, sj, out, (hipStream_t)stream));
  return BORNVI_OK;
}

size_t bornvi_stein_quadform_workspace_bytes(bornvi_handle h, int n, int B) {
  (void)h;
  if (n < 1 || n > 30) return 0;
  size_t b = align_up(quadform_partials(1ll << n) * sizeof(double), 256);
  // the batched (matrix-core) form computes ksd2 from Y: room for Y when the caller does not ask for it
  if (B > 1 && quadform_batched_supported(n, B)) b += align_up((size_t)B * ((size_t)8 << n), 256);
  return b;
}

int bornvi_stein_quadform_rows(bornvi_handle h, int n, const double* K_rows, long long row_begin, long long row_end,
                               const double* q, double* y_rows, double* ksd2_partial, void* workspace,
                               size_t workspace_bytes, bornvi_stream stream) {
  if (!h) return BORNVI_ERR_INVALID;
  if (!q || !ksd2_partial || (!K_rows && row_end > row_begin)) return fail(h, BORNVI_ERR_INVALID, "bad argument");
  if (!valid_n_for_dense(n)) return fail(h, BORNVI_ERR_UNSUPPORTED, "dense Gram supports 1 <= n <= 17");
  if (row_begin < 0 || row_end < row_begin || row_end > (1ll << n)) return fail(h, BORNVI_ERR_INVALID, "row range out of bounds");
  if (!workspace || workspace_bytes < bornvi_stein_quadform_workspace_bytes(h, n, 1))
    return fail(h, BORNVI_ERR_WORKSPACE, "workspace too small");
  DEVICE_SCOPE(h);
  HIPCHK(h, launch_quadform(n, K_rows, row_begin, row_end, q, y_rows, ksd2_partial, (double*)workspace, (hipStream_t)stream));
  return BORNVI_OK;
}

int bornvi_stein_quadform_ld(bornvi_handle h, int n, const double* K, long long ld, const double* Q, int B, double* ksd2,
                             double* Y, void* workspace, size_t workspace_bytes, bornvi_stream stream) {
  if (!h) return BORNVI_ERR_INVALID;
  if (!K || !Q || !ksd2 || B < 0) return fail(h, BORNVI_ERR_INVALID, "bad argument");
  if (!valid_n_for_dense(n)) return fail(h, BORNVI_ERR_UNSUPPORTED, "dense Gram supports 1 <= n <= 17");
  if (!valid_ld(n, ld)) return fail(h, BORNVI_ERR_INVALID, "leading dimension must be even, >= 2^n and <= 2^n + 4096");
  if (!workspace || workspace_bytes < bornvi_stein_quadform_workspace_bytes(h, n, B))
    return fail(h, BORNVI_ERR_WORKSPACE, "workspace too small");
  DEVICE_SCOPE(h);
  const long long N = 1ll << n;
  if (quadform_batched_supported(n, B) && h->batched_quadform) {
    // one pass over K on the matrix cores (kernels_batched.hip) instead of B passes of the HBM-bound GEMV
    double* Yw = Y ? Y : (double*)((char*)workspace + align_up(quadform_partials(N) * sizeof(double), 256));
    HIPCHK(h, launch_quadform_batched(n, K, ld, Q, B, Yw, ksd2, (hipStream_t)stream));
    return BORNVI_OK;
  }
  if (ld != N) return fail(h, BORNVI_ERR_UNSUPPORTED, "a padded K needs the batched form (n >= 8, B >= 2, option batched_quadform = 1); "
                                                      "B = 1: bornvi_stein_quadform_sym_ld");
  for (int b = 0; b < B; ++b)
    HIPCHK(h, launch_quadform(n, K, 0, N, Q + b * N, Y ? Y + b * N : nullptr, ksd2 + b, (double*)workspace, (hipStream_t)stream));
  return BORNVI_OK;
}

int bornvi_stein_quadform(bornvi_handle h, int n, const double* K, const double* Q, int B, double* ksd2, double* Y,
                          void* workspace, size_t workspace_bytes, bornvi_stream stream) {
  if (n < 0 || n > 40) return h ? fail(h, BORNVI_ERR_UNSUPPORTED, "dense Gram supports 1 <= n <= 17") : BORNVI_ERR_INVALID;
  return bornvi_stein_quadform_ld(h, n, K, 1ll << n, Q, B, ksd2, Y, workspace, workspace_bytes, stream);
}

size_t bornvi_stein_quadform_sym_workspace_bytes(bornvi_handle h, int n) {
  (void)h;
  if (!valid_n_for_dense(n)) return 0;
  return align_up(quadform_sym_workspace_doubles(n) * sizeof(double), 256);
}

int bornvi_stein_quadform_sym_ld(bornvi_handle h, int n, const double* K, long long ld, const double* q, double* ksd2,
                                 double* y, void* workspace, size_t workspace_bytes, bornvi_stream stream) {
  if (!h) return BORNVI_ERR_INVALID;
  if (!K || !q || !ksd2) return fail(h, BORNVI_ERR_INVALID, "null pointer");
  if (!valid_n_for_dense(n)) return fail(h, BORNVI_ERR_UNSUPPORTED, "dense Gram supports 1 <= n <= 17");
  if (!valid_ld(n, ld)) return fail(h, BORNVI_ERR_INVALID, "leading dimension must be even, >= 2^n and <= 2^n + 4096");
  if (!workspace || workspace_bytes < bornvi_stein_quadform_sym_workspace_bytes(h, n) || ((uintptr_t)workspace & 15))
    return fail(h, BORNVI_ERR_WORKSPACE, "workspace too small or not 16-byte aligned");
  if ((uintptr_t)K & 15) return fail(h, BORNVI_ERR_INVALID, "K must be 16-byte aligned");
  if (n < quadform_sym_min_n() && ld != (1ll << n))
    return fail(h, BORNVI_ERR_UNSUPPORTED, "a padded K needs at least two 256-row bands (n >= 9); smaller matrices are contracted dense");
  DEVICE_SCOPE(h);
  HIPCHK(h, launch_quadform_sym(n, K, ld, q, y, ksd2, (double*)workspace, (hipStream_t)stream));
  return BORNVI_OK;
}

int bornvi_stein_quadform_sym(bornvi_handle h, int n, const double* K, const double* q, double* ksd2, double* y,
                              void* workspace, size_t workspace_bytes, bornvi_stream stream) {
  if (n < 0 || n > 40) return h ? fail(h, BORNVI_ERR_UNSUPPORTED, "dense Gram supports 1 <= n <= 17") : BORNVI_ERR_INVALID;
  return bornvi_stein_quadform_sym_ld(h, n, K, 1ll << n, q, ksd2, y, workspace, workspace_bytes, stream);
}

int bornvi_stein_sym_strip_rows(void) { return quadform_sym_rows_per_strip(); }

int bornvi_stein_quadform_sym_pairs_ld(bornvi_handle h, int n, const double* K_lo, const double* K_hi, long long ld,
                                       long long pair_begin, long long pair_end, const double* q, double* ksd2_partial,
                                       double* y_partial, void* workspace, size_t workspace_bytes, bornvi_stream stream) {
  if (!h) return BORNVI_ERR_INVALID;
  if (!q || !ksd2_partial || !y_partial) return fail(h, BORNVI_ERR_INVALID, "null pointer");
  if (!valid_n_for_dense(n)) return fail(h, BORNVI_ERR_UNSUPPORTED, "dense Gram supports 1 <= n <= 17");
  if (!valid_ld(n, ld)) return fail(h, BORNVI_ERR_INVALID, "leading dimension must be even, >= 2^n and <= 2^n + 4096");
  if (n < quadform_sym_min_n()) return fail(h, BORNVI_ERR_UNSUPPORTED, "the strip-pair shard needs at least two bands of rows (n >= 9)");
  const long long N = 1ll << n, R = quadform_sym_rows_per_strip();
  const long long ns = N / R, npairs = (ns + 1) / 2;
  if (pair_begin < 0 || pair_end < pair_begin || pair_end > npairs) return fail(h, BORNVI_ERR_INVALID, "strip-pair range out of bounds");
  if (pair_end > pair_begin && (!K_lo || !K_hi)) return fail(h, BORNVI_ERR_INVALID, "null pointer");
  if (((uintptr_t)K_lo | (uintptr_t)K_hi) & 15) return fail(h, BORNVI_ERR_INVALID, "K blocks must be 16-byte aligned");
  if (!workspace || workspace_bytes < bornvi_stein_quadform_sym_workspace_bytes(h, n) || ((uintptr_t)workspace & 15))
    return fail(h, BORNVI_ERR_WORKSPACE, "workspace too small or not 16-byte aligned");
  DEVICE_SCOPE(h);
  HIPCHK(h, launch_quadform_sym_pairs(n, K_lo, K_hi, ld, pair_begin, pair_end, q, y_partial, ksd2_partial, (double*)workspace,
                                      (hipStream_t)stream));
  return BORNVI_OK;
}

int bornvi_stein_quadform_sym_pairs(bornvi_handle h, int n, const double* K_lo, const double* K_hi, long long pair_begin,
                                    long long pair_end, const double* q, double* ksd2_partial, double* y_partial,
                                    void* workspace, size_t workspace_bytes, bornvi_stream stream) {
  if (n < 0 || n > 40) return h ? fail(h, BORNVI_ERR_UNSUPPORTED, "dense Gram supports 1 <= n <= 17") : BORNVI_ERR_INVALID;
  return bornvi_stein_quadform_sym_pairs_ld(h, n, K_lo, K_hi, 1ll << n, pair_begin, pair_end, q, ksd2_partial, y_partial,
                                            workspace, workspace_bytes, stream);
}

size_t bornvi_stein_matvec_kron_workspace_bytes(bornvi_handle h, int n) {
  if (!h || n < 1 || n > 30) return 0;
  DevPlan* dp = nullptr;
  if (get_plan(h, -1, n, 0, &dp)) return 0;
  const size_t npk = (size_t)(n + 2) / 2;
  const size_t st = align_up(npk * ((size_t)16 << n), 256);
  const size_t nbuf = dp->plan.n_passes >= 3 ? 3 : (dp->plan.n_passes == 2 ? 2 : 1);
  return 256 + nbuf * st + align_up(kron_partials(n) * sizeof(double), 256);
}

int bornvi_stein_matvec_kron(bornvi_handle h, int n, double length_scale, const double* S, const double* q, double* y,
                             double* ksd2, void* workspace, size_t workspace_bytes, bornvi_stream stream) {
  if (!h) return BORNVI_ERR_INVALID;
  if (!S || !q || !ksd2) return fail(h, BORNVI_ERR_INVALID, "null pointer");
  if (!(length_scale > 0.0)) return fail(h, BORNVI_ERR_INVALID, "length_scale must be positive");
  const size_t need = bornvi_stein_matvec_kron_workspace_bytes(h, n);
  if (!need) return h->err.empty() ? fail(h, BORNVI_ERR_UNSUPPORTED, "unsupported n") : BORNVI_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < need) return fail(h, BORNVI_ERR_WORKSPACE, "workspace too small");
  DevPlan* dp = nullptr;
  int rc = get_plan(h, -1, n, 0, &dp);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int npk = (n + 2) / 2;
  const size_t stb = align_up((size_t)npk * ((size_t)16 << n), 256);
  char* base = (char*)workspace;
  double* gate = (double*)base;
  void* X = base + 256;
  void* A = base + 256 + stb;
  void* Bf = base + 256 + 2 * stb;
  const int np = dp->plan.n_passes;
  double* partials = (double*)(base + 256 + (np >= 3 ? 3 : (np == 2 ? 2 : 1)) * stb);
  DEVICE_SCOPE(h);
  HIPCHK(h, launch_kron_pack(n, length_scale, S, q, (double*)X, gate, st));
  // (the 8-amplitude kernel reads pivot-normalised records; M = [[1, a], [a, 1]] has pivot 1: no scale to carry)
  if (dp->d_compact) HIPCHK(h, launch_normalise_gates(gate, 1, st));
  // X -> (A -> B -> A ...) -> X : the last pass writes back into X (dead after pass 0; with a single
  // pass each workgroup owns a whole state and reads it completely before writing).
  rc = run_passes(h, dp, npk, X, A, Bf, X, nullptr, gate, 0, st);
  if (rc) return rc;
  HIPCHK(h, launch_kron_combine(n, S, q, (const double*)X, y, partials, ksd2, st));
  return BORNVI_OK;
}

int bornvi_ksd_grad_finish(bornvi_handle h, int n, const double* shifted, int n_shift, const double* y,
                           const double* ksd2, double* loss_out, double* dLdq_out, double* grad, bornvi_stream stream) {
  if (!h) return BORNVI_ERR_INVALID;
  if (!y || !ksd2 || n < 1 || n > 30 || n_shift < 0 || (n_shift > 0 && (!shifted || !grad)))
    return fail(h, BORNVI_ERR_INVALID, "bad argument");
  DEVICE_SCOPE(h);
  hipStream_t st = (hipStream_t)stream;
  // (the loss alone rides on the dot kernel -- one launch less for the latency-bound sizes; dL/dq needs its own pass)
  const bool loss_by_dot = loss_out && !dLdq_out && n_shift > 0;
  if (dLdq_out || (loss_out && !loss_by_dot)) HIPCHK(h, launch_dldq(y, ksd2, n, dLdq_out, loss_out, st));
  HIPCHK(h, launch_shift_dot(shifted, n_shift, y, ksd2, n, grad, loss_by_dot ? loss_out : nullptr, st));
  return BORNVI_OK;
}

int bornvi_clip_cast_grad(bornvi_handle h, int P, const double* grad64, double max_norm, float* grad32,
                          float* total_norm, bornvi_stream stream) {
  if (!h) return BORNVI_ERR_INVALID;
  if (P < 0 || (P > 0 && (!grad64 || !grad32)) || !total_norm || !(max_norm >= 0.0)) return fail(h, BORNVI_ERR_INVALID, "bad argument");
  DEVICE_SCOPE(h);
  HIPCHK(h, launch_clip_cast(grad64, P, max_norm, grad32, total_norm, nullptr, nullptr, (hipStream_t)stream));
  return BORNVI_OK;
}

int bornvi_clip_cast_grad_guard(bornvi_handle h, int P, const double* grad64, double max_norm, const double* loss,
                                float* grad32, float* total_norm, float* found_inf, bornvi_stream stream) {
  if (!h) return BORNVI_ERR_INVALID;
  if (P < 0 || (P > 0 && (!grad64 || !grad32)) || !total_norm || !loss || !found_inf || !(max_norm >= 0.0))
    return fail(h, BORNVI_ERR_INVALID, "bad argument");
  DEVICE_SCOPE(h);
  HIPCHK(h, launch_clip_cast(grad64, P, max_norm, grad32, total_norm, loss, found_inf, (hipStream_t)stream));
  return BORNVI_OK;
}

int bornvi_clip_adam_step(bornvi_handle h, int P, const double* grad64, double max_norm, const double* loss, float* theta,
                          float* grad32, double* theta64, float* exp_avg, float* exp_avg_sq, int* counters,
                          const double* lr_table, int n_lr, double beta1, double beta2, double eps, float* total_norm,
                          double* loss_history, float* norm_history, bornvi_stream stream) {
  if (!h) return BORNVI_ERR_INVALID;
  if (P < 1 || !grad64 || !loss || !theta || !grad32 || !theta64 || !exp_avg || !exp_avg_sq || !counters || !lr_table ||
      n_lr < 1 || !total_norm || !(max_norm >= 0.0))
    return fail(h, BORNVI_ERR_INVALID, "bad argument");
  if (!(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0))
    return fail(h, BORNVI_ERR_INVALID, "Adam needs 0 <= beta < 1 and eps >= 0");
  DEVICE_SCOPE(h);
  HIPCHK(h, launch_clip_adam(grad64, P, max_norm, loss, theta, grad32, theta64, exp_avg, exp_avg_sq, counters, lr_table, n_lr,
                             beta1, beta2, eps, total_norm, loss_history, norm_history, (hipStream_t)stream));
  return BORNVI_OK;
}

size_t bornvi_adjoint_workspace_bytes(bornvi_handle h, int ansatz, int n, int layers) {
  if (!h) return 0;
  AdjPlan* ap = nullptr;
  if (get_adj_plan(h, ansatz, n, layers, &ap)) return 0;
  const size_t st = align_up((size_t)16 << n, 256);
  return 256 + 4 * st + align_up((ap->rots.size() + 1) * (size_t)adjoint_workgroups(n) * 4 * sizeof(double), 256);
}

int bornvi_adjoint_state(bornvi_handle h, int ansatz, int n, int layers, const double* theta, double* state, double* probs,
                         void* workspace, size_t workspace_bytes, bornvi_stream stream) {
  if (!h) return BORNVI_ERR_INVALID;
  AdjPlan* ap = nullptr;
  int rc = get_adj_plan(h, ansatz, n, layers, &ap);
  if (rc) return rc;
  if (!state || (!theta && ap->n_params > 0)) return fail(h, BORNVI_ERR_INVALID, "null pointer");
  if (!workspace || workspace_bytes < bornvi_adjoint_workspace_bytes(h, ansatz, n, layers)) return fail(h, BORNVI_ERR_WORKSPACE, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  DEVICE_SCOPE(h);
  double* other = (double*)((char*)workspace + 256);
  const int nent = (int)ap->ents_fwd.size();
  double* cur = (nent % 2 == 0) ? state : other;            // every entangler block swaps: end in `state`
  double* alt = (nent % 2 == 0) ? other : state;
  HIPCHK(h, launch_adj_init(cur, n, st));
  for (const AdjPlan::Step& s : ap->steps) {
    if (s.is_rot) {
      HIPCHK(h, launch_adj_rot_forward(cur, n, ap->rots[(size_t)s.index], theta, st));
    } else {
      HIPCHK(h, launch_adj_entangle(cur, alt, nullptr, nullptr, n, ap->ents_fwd[(size_t)s.index], 0, st));
      std::swap(cur, alt);
    }
  }
  if (probs) HIPCHK(h, launch_born_probs(state, probs, n, 1, st));
  return BORNVI_OK;
}

int bornvi_adjoint_vjp(bornvi_handle h, int ansatz, int n, int layers, const double* theta, const double* state,
                       const double* dLdq, double* grad, void* workspace, size_t workspace_bytes, bornvi_stream stream) {
  if (!h) return BORNVI_ERR_INVALID;
  AdjPlan* ap = nullptr;
  int rc = get_adj_plan(h, ansatz, n, layers, &ap);
  if (rc) return rc;
  if (ap->n_params == 0) return BORNVI_OK;
  if (!state || !dLdq || !grad || !theta) return fail(h, BORNVI_ERR_INVALID, "null pointer");
  if (!workspace || workspace_bytes < bornvi_adjoint_workspace_bytes(h, ansatz, n, layers)) return fail(h, BORNVI_ERR_WORKSPACE, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  DEVICE_SCOPE(h);
  const size_t sb = align_up((size_t)16 << n, 256);
  char* base = (char*)workspace + 256;
  double* phi = (double*)base;
  double* phi2 = (double*)(base + sb);
  double* lam = (double*)(base + 2 * sb);
  double* lam2 = (double*)(base + 3 * sb);
  double* partials = (double*)(base + 4 * sb);
  const int nwg = adjoint_workgroups(n);
  HIPCHK(h, hipMemcpyAsync(phi, state, (size_t)16 << n, hipMemcpyDeviceToDevice, st));
  HIPCHK(h, launch_adj_lambda(state, dLdq, lam, n, st));
  for (auto it = ap->steps.rbegin(); it != ap->steps.rend(); ++it) {
    if (it->is_rot) {
      HIPCHK(h, launch_adj_rot_backward(phi, lam, n, ap->rots[(size_t)it->index], theta, partials + (size_t)it->index * nwg * 4, st));
    } else {
      HIPCHK(h, launch_adj_entangle(phi, phi2, lam, lam2, n, ap->ents_bwd[(size_t)it->index], 1, st));
      std::swap(phi, phi2);
      std::swap(lam, lam2);
    }
  }
  // (a parameter no gate carries keeps gradient 0; every parameter sits in exactly one slot -- build_adj_plan checks it)
  HIPCHK(h, hipMemsetAsync(grad, 0, (size_t)(ap->n_params > 0 ? ap->n_params : 0) * sizeof(double), st));
  HIPCHK(h, launch_adj_reduce(partials, ap->d_slot_param, 4 * (int)ap->rots.size(), nwg, grad, st));
  return BORNVI_OK;
}

int bornvi_debug_circuit_stamps(bornvi_handle h, unsigned long long* out16) {
  if (!h || !out16) return BORNVI_ERR_INVALID;
  DEVICE_SCOPE(h);
  { int rc_ = sync_uploads(h); if (rc_) return rc_; }
  HIPCHK(h, read_circuit_stamps(out16));
  return BORNVI_OK;
}

long long bornvi_plan_describe(int ansatz, int n, int layers, int tile_bits, uint32_t* out, size_t cap_words) {
  PlanOptions opt;
  opt.read_map = (tile_bits & 0x100) ? 1 : 0;    // (bit 8 of tile_bits: the planner's read_map option)
  opt.r = (tile_bits & 0x200) ? 3 : 4;           // (bit 9: 3 register wires, the plan circuit_pass_r3_kernel runs)
  if (opt.r == 4) opt.max_threads = 512;
  tile_bits &= 0xff;
  if (tile_bits > 0) { opt.kmax = tile_bits; opt.kmulti = tile_bits; }
  Plan p;
  std::string msg;
  const bool ok = (ansatz == -1) ? make_kron_plan(n, opt, p, msg) : make_plan(ansatz, n, layers, opt, p, msg);
  if (!ok) return -1;
  if (out) {
    const size_t c = p.words.size() < cap_words ? p.words.size() : cap_words;
    std::memcpy(out, p.words.data(), c * sizeof(uint32_t));
  }
  return (long long)p.words.size();
}

int bornvi_plan_param_first_pass(int ansatz, int n, int layers, int tile_bits, int* out, int cap) {
  PlanOptions opt;
  opt.r = (tile_bits & 0x200) ? 3 : 4;
  opt.read_map = (tile_bits & 0x100) ? 1 : 0;
  if (opt.r == 4) opt.max_threads = 512;
  tile_bits &= 0xff;
  if (tile_bits > 0) { opt.kmax = tile_bits; opt.kmulti = tile_bits; }
  Plan p;
  std::string msg;
  if (ansatz < 0 || !make_plan(ansatz, n, layers, opt, p, msg)) return -1;
  if (out)
    for (int q = 0; q < p.n_params && q < cap; ++q) out[q] = p.param_first_pass[q];
  return p.n_params;
}

long long bornvi_plan_fast_describe(int ansatz, int n, int layers, int tile_bits, uint32_t* out, size_t cap_words,
                                    uint32_t* pass_off_out, int cap_passes) {
  PlanOptions opt;
  opt.read_map = (tile_bits & 0x100) ? 1 : 0;
  opt.r = (tile_bits & 0x200) ? 3 : 4;
  if (opt.r == 4) opt.max_threads = 512;
  tile_bits &= 0xff;
  if (tile_bits > 0) { opt.kmax = tile_bits; opt.kmulti = tile_bits; }
  Plan p;
  std::string msg;
  const bool ok = (ansatz == -1) ? make_kron_plan(n, opt, p, msg) : make_plan(ansatz, n, layers, opt, p, msg);
  if (!ok) return -1;
  FastTables ft;
  if (!build_fast_tables(p, FAST_TABLE_MAX_BYTES, ft)) return 0;
  if (out) {
    const size_t c = ft.words.size() < cap_words ? ft.words.size() : cap_words;
    std::memcpy(out, ft.words.data(), c * sizeof(uint32_t));
  }
  if (pass_off_out)
    for (int i = 0; i < cap_passes && i < (int)ft.pass_off.size(); ++i) pass_off_out[i] = ft.pass_off[i];
  return (long long)ft.words.size();
}

// Compact tables of the 3-register-wire plan (plan.hpp: CompactTables): returns the number of words (0: not eligible, -1:
// no plan); every word was checked against its point evaluation by build_compact_tables.
long long bornvi_plan_compact_describe(int ansatz, int n, int layers, int tile_bits, uint32_t* out, size_t cap_words,
                                       uint32_t* pass_off_out, int cap_passes) {
  PlanOptions opt;
  opt.read_map = (tile_bits & 0x100) ? 1 : 0;
  opt.r = 3;
  tile_bits &= 0xff;
  if (tile_bits > 0) { opt.kmax = tile_bits; opt.kmulti = tile_bits; }
  Plan p;
  std::string msg;
  const bool ok = (ansatz == -1) ? make_kron_plan(n, opt, p, msg) : make_plan(ansatz, n, layers, opt, p, msg);
  if (!ok) return -1;
  CompactTables ct;
  if (!build_compact_tables(p, ct, msg) || ct.lds_bytes(p.k) > MAX_LDS_BYTES) return 0;
  if (out) {
    const size_t c = ct.words.size() < cap_words ? ct.words.size() : cap_words;
    std::memcpy(out, ct.words.data(), c * sizeof(uint32_t));
  }
  if (pass_off_out)
    for (int i = 0; i < cap_passes && i < (int)ct.pass_off.size(); ++i) pass_off_out[i] = ct.pass_off[i];
  return (long long)ct.words.size();
}

}  // extern "C"
