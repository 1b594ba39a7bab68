// Circuit execution planner -- see plan.hpp for the model.
#include "plan.hpp"
#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <array>
#include <cstring>
#include <utility>

namespace bornvi {

int num_params(int ansatz, int n, int layers) {
  // quantum_born_machine.py:31-38
  if (ansatz == 0 || ansatz == 1) return layers * 3 * n;
  if (ansatz == 2) return layers * 2 * n;
  return -1;
}

bool build_gate_list(int ansatz, int n, int layers, std::vector<Gate>& g) {
  g.clear();
  int p = 0;
  auto rot = [&](int kind, int w) { g.push_back({kind, w, -1, p++}); };
  if (ansatz == 0) {  // hardware_efficient, quantum_born_machine.py:58-87
    for (int i = 0; i < n; ++i) g.push_back({G_H, i, -1, -1});
    for (int l = 0; l < layers; ++l) {
      for (int i = 0; i < n; ++i) { rot(G_RX, i); rot(G_RY, i); rot(G_RZ, i); }
      if (n > 1) {
        for (int i = 0; i + 1 < n; ++i) g.push_back({G_CNOT, i, i + 1, -1});
        if (n > 2) g.push_back({G_CNOT, n - 1, 0, -1});
        if (l % 2 == 0 && n > 2)
          for (int i = 0; i < n - 2; i += 2) g.push_back({G_CZ, i, i + 2, -1});
      }
    }
  } else if (ansatz == 1) {  // all_to_all, quantum_born_machine.py:90-111
    for (int i = 0; i < n; ++i) g.push_back({G_H, i, -1, -1});
    for (int l = 0; l < layers; ++l) {
      for (int i = 0; i < n; ++i) { rot(G_RX, i); rot(G_RY, i); rot(G_RZ, i); }
      if (n > 1)
        for (int i = 0; i < n; ++i)
          for (int j = i + 1; j < n; ++j) g.push_back({G_CZ, i, j, -1});
    }
  } else if (ansatz == 2) {  // basic, quantum_born_machine.py:114-128
    for (int l = 0; l < layers; ++l) {
      for (int i = 0; i < n; ++i) { rot(G_RY, i); rot(G_RZ, i); }
      if (n > 1) {
        for (int i = 0; i + 1 < n; ++i) g.push_back({G_CNOT, i, i + 1, -1});
        if (n > 2) g.push_back({G_CNOT, n - 1, 0, -1});
      }
    }
  } else {
    return false;
  }
  return true;
}

namespace {

enum { K_U1 = 0, K_CX = 1, K_CZ = 2 };
struct Op { int kind, a, b, idx; };  // U1: a = wire, idx = fused gate; CX: a = control, b = target; CZ: a, b
using Fused = std::array<uint32_t, FUSED_WORDS>;

void fuse_gates(const std::vector<Gate>& gates, int n, std::vector<Op>& ops, std::vector<Fused>& fused) {
  std::vector<int> open(n, -1);
  for (const Gate& g : gates) {
    if (g.kind <= G_RZ) {
      int f = open[g.w0];
      if (f < 0 || fused[f][1] >= (uint32_t)FUSED_MAX_ELEMS) {
        Fused nf; nf.fill(0xffffffffu);
        nf[0] = (uint32_t)g.w0; nf[1] = 0;
        fused.push_back(nf);
        f = (int)fused.size() - 1;
        open[g.w0] = f;
        ops.push_back({K_U1, g.w0, -1, f});
      }
      uint32_t e = fused[f][1]++;
      fused[f][2 + 2 * e] = (uint32_t)g.kind;
      fused[f][3 + 2 * e] = (uint32_t)g.param;  // -1 -> 0xffffffff
    } else {
      open[g.w0] = open[g.w1] = -1;
      ops.push_back({g.kind == G_CNOT ? K_CX : K_CZ, g.w0, g.w1, -1});
    }
  }
}

inline int op_target(const Op& o) { return o.kind == K_U1 ? o.a : (o.kind == K_CX ? o.b : -1); }

// Greedy selection in program order: an op runs if none of its wires is blocked and its
// non-diagonal target is (or can become) one of at most `cap` "near" wires.  Anything that
// cannot run blocks its wires for the rest of the scan (ops on disjoint wires commute).
void greedy_select(const std::vector<Op>& ops, const std::vector<int>& pool, int n, int cap,
                   std::vector<int>& sel, std::vector<int>& rest, std::vector<int>& targets) {
  std::vector<char> blocked(n, 0), in_t(n, 0);
  sel.clear(); rest.clear(); targets.clear();
  for (int idx : pool) {
    const Op& o = ops[idx];
    bool blk = blocked[o.a] || (o.b >= 0 && blocked[o.b]);
    int t = op_target(o);
    if (!blk && t >= 0 && !in_t[t]) {
      if ((int)targets.size() < cap) { in_t[t] = 1; targets.push_back(t); }
      else blk = true;
    }
    if (blk) {
      blocked[o.a] = 1;
      if (o.b >= 0) blocked[o.b] = 1;
      rest.push_back(idx);
    } else {
      sel.push_back(idx);
    }
  }
}

// Same scan with a FIXED set of allowed target wires (a candidate local set of a pass).
// Returns the number of locality-bound ops (U1, CNOT) it executes.
int fixed_select(const std::vector<Op>& ops, const std::vector<int>& pool, int n, const std::vector<char>& allowed,
                 std::vector<int>& sel, std::vector<int>& rest, std::vector<int>& targets) {
  std::vector<char> blocked(n, 0), in_t(n, 0);
  sel.clear(); rest.clear(); targets.clear();
  int bound_ops = 0;
  for (int idx : pool) {
    const Op& o = ops[idx];
    bool blk = blocked[o.a] || (o.b >= 0 && blocked[o.b]);
    const int t = op_target(o);
    if (!blk && t >= 0 && !allowed[t]) blk = true;
    if (blk) {
      blocked[o.a] = 1;
      if (o.b >= 0) blocked[o.b] = 1;
      rest.push_back(idx);
    } else {
      if (t >= 0) { ++bound_ops; if (!in_t[t]) { in_t[t] = 1; targets.push_back(t); } }
      sel.push_back(idx);
    }
  }
  return bound_ops;
}

// Pass selection: the program-order greedy set and every cyclic window of k consecutive wires are tried;
// the candidate that lets this pass plus the best following pass execute the most locality-bound ops wins
// (one step of look-ahead).  Fewer passes = fewer HBM round trips of the whole batch of states.
void choose_pass(const std::vector<Op>& ops, const std::vector<int>& pool, int n, int k, std::vector<int>& sel,
                 std::vector<int>& rest, std::vector<int>& targets) {
  std::vector<std::vector<char>> cands;
  {
    std::vector<int> s, r, t;
    greedy_select(ops, pool, n, k, s, r, t);
    std::vector<char> a(n, 0);
    for (int w : t) a[w] = 1;
    int cnt = (int)t.size();
    for (int w = n - 1; w >= 0 && cnt < k; --w) if (!a[w]) { a[w] = 1; ++cnt; }
    cands.push_back(a);
  }
  if (k < n)
    for (int s0 = 0; s0 < n; ++s0) {
      std::vector<char> a(n, 0);
      for (int j = 0; j < k; ++j) a[(s0 + j) % n] = 1;
      cands.push_back(a);
    }
  long best_score = -1;
  std::vector<int> s1, r1, t1, s2, r2, t2;
  for (const auto& a : cands) {
    const int here = fixed_select(ops, pool, n, a, s1, r1, t1);
    if (s1.empty()) continue;
    int next_best = 0;
    if (!r1.empty() && k < n)
      for (const auto& a2 : cands) {   // (the greedy candidate of the NEXT pool is approximated by the windows)
        const int nb = fixed_select(ops, r1, n, a2, s2, r2, t2);
        if (nb > next_best) next_best = nb;
      }
    const long score = (long)(here + next_best) * 4096 + (long)s1.size();
    if (score > best_score) { best_score = score; sel = s1; rest = r1; targets = t1; }
  }
  if (best_score < 0) greedy_select(ops, pool, n, k, sel, rest, targets);
}

// Stage selection.  A stage runs its ops in five global phases -- 0: CNOTs (folded into the LDS read
// address), 1: CZs, 2: at most one fused U per register wire, 3: CNOTs (folded into the write address),
// 4: CZs -- so an op is accepted in the earliest phase its kind allows that is not before the phase of
// the last op accepted on any of its wires (ops on disjoint wires commute).
struct StageSel {
  std::vector<int> pre_cx, pre_cz, us, post_cx, post_cz, targets;
  size_t count() const { return pre_cx.size() + pre_cz.size() + us.size() + post_cx.size() + post_cz.size(); }
};

// defer_cx3: a CNOT that could only run in phase 3 AND would claim a new register wire is left for the next stage,
// where it is a phase-0 CNOT (folded into the read map: free) -- the register wires then go to fused U's.
void stage_select(const std::vector<Op>& ops, const std::vector<int>& pool, int n, int cap, StageSel& S,
                  std::vector<int>& rest, bool defer_cx3 = false, bool read_map = false) {
  std::vector<char> blocked(n, 0), in_t(n, 0), has_u(n, 0);
  std::vector<int> wphase(n, 0);
  rest.clear();
  for (int idx : pool) {
    const Op& o = ops[idx];
    bool blk = blocked[o.a] || (o.b >= 0 && blocked[o.b]);
    int ph = -1;
    if (!blk) {
      const int mx = std::max(wphase[o.a], o.b >= 0 ? wphase[o.b] : 0);
      if (o.kind == K_U1) ph = (mx <= 1 && !has_u[o.a]) ? 2 : -1;
      else if (o.kind == K_CX) ph = mx <= 0 ? 0 : (mx <= 3 ? 3 : -1);
      else ph = mx <= 1 ? 1 : (mx <= 4 ? 4 : -1);
      if (ph < 0) blk = true;
    }
    const int t = op_target(o);
    // (a phase-0 CNOT is folded into the stage's READ map, which may move any local bit: its target need not be one of
    // the `cap` register wires -- only a fused U and a phase-3 CNOT, whose write-back must stay inside the thread's own
    // group, claim one)
    if (!blk && t >= 0 && !in_t[t] && !(read_map && o.kind == K_CX && ph == 0)) {
      if (defer_cx3 && o.kind == K_CX) blk = true;
      else if ((int)S.targets.size() < cap) { in_t[t] = 1; S.targets.push_back(t); }
      else blk = true;
    }
    if (blk) {
      blocked[o.a] = 1;
      if (o.b >= 0) blocked[o.b] = 1;
      rest.push_back(idx);
      continue;
    }
    wphase[o.a] = std::max(wphase[o.a], ph);
    if (o.b >= 0) wphase[o.b] = std::max(wphase[o.b], ph);
    switch (ph) {
      case 0: S.pre_cx.push_back(idx); break;
      case 1: S.pre_cz.push_back(idx); break;
      case 2: S.us.push_back(idx); has_u[o.a] = 1; break;
      case 3: S.post_cx.push_back(idx); break;
      default: S.post_cz.push_back(idx); break;
    }
  }
}

// Order LDS bit positions so that consecutive lanes are bank-conflict free under the
// XOR-fold swizzle (phys low nibble = xor of all nibbles of the index): lane bits 0..2 go to
// positions with residues {0,1,2} mod 4, lane bit 3 to residue 3, lane bit 4 to the residue of
// lane bit 0 or 1 (ds_read_b128 services lanes {0-3,12-15,20-27} together).
std::vector<int> order_for_banks(std::vector<int> pos) {
  std::sort(pos.begin(), pos.end());
  std::vector<int> out;
  std::vector<char> used(pos.size(), 0);
  bool res_used[4] = {false, false, false, false};
  auto take = [&](auto pred) -> bool {
    for (size_t i = 0; i < pos.size(); ++i)
      if (!used[i] && pred(pos[i])) { used[i] = 1; out.push_back(pos[i]); res_used[pos[i] & 3] = true; return true; }
    return false;
  };
  for (int s = 0; s < 3 && out.size() < pos.size(); ++s)
    if (!take([&](int p) { return (p & 3) != 3 && !res_used[p & 3]; }))
      take([&](int) { return true; });
  if (out.size() < pos.size())
    if (!take([&](int p) { return (p & 3) == 3; })) take([&](int) { return true; });
  if (out.size() < pos.size() && out.size() >= 2) {
    int r0 = out[0] & 3, r1 = out[1] & 3;
    if (!take([&](int p) { return (p & 3) == r0 || (p & 3) == r1; })) take([&](int) { return true; });
  }
  for (size_t i = 0; i < pos.size(); ++i)
    if (!used[i]) out.push_back(pos[i]);
  return out;
}

// lead / core / trail split of a pass's ops: CNOTs at the head / tail that commute past everything before / after
// them in the pass are folded into the tile load / store addressing; the rest (core) runs in stages
void split_pass_ops(const std::vector<Op>& ops, const std::vector<int>& pass_ops, int n, bool is_init,
                    std::vector<int>& lead, std::vector<int>& core, std::vector<int>& trail) {
  lead.clear(); core.clear(); trail.clear();
  std::vector<char> touched(n, 0);
  std::vector<int> mid;
  for (int idx : pass_ops) {
    const Op& o = ops[idx];
    if (!is_init && o.kind == K_CX && !touched[o.a] && !touched[o.b]) { lead.push_back(idx); continue; }
    touched[o.a] = 1; if (o.b >= 0) touched[o.b] = 1;
    mid.push_back(idx);
  }
  std::fill(touched.begin(), touched.end(), 0);
  std::vector<int> trail_rev, core_rev;
  for (size_t q = mid.size(); q-- > 0;) {
    const Op& o = ops[mid[q]];
    if (o.kind == K_CX && !touched[o.a] && !touched[o.b]) { trail_rev.push_back(mid[q]); continue; }
    touched[o.a] = 1; if (o.b >= 0) touched[o.b] = 1;
    core_rev.push_back(mid[q]);
  }
  core.assign(core_rev.rbegin(), core_rev.rend());
  trail.assign(trail_rev.rbegin(), trail_rev.rend());
}

// target wires of the first and of the last stage of a pass (they depend on the ops only, not on the layout)
// number of stages the core ops of a pass need under a stage-selection policy
int count_stages(const std::vector<Op>& ops, const std::vector<int>& core, int n, int r, bool defer_cx3, bool read_map) {
  std::vector<int> pool = core, rest;
  int cnt = 0;
  while (!pool.empty()) {
    StageSel sel;
    stage_select(ops, pool, n, r, sel, rest, defer_cx3, read_map);
    if (sel.count() == 0) return 1 << 20;
    ++cnt;
    pool = rest;
  }
  return cnt;
}
// the policy with fewer stages (ties: the program-order greedy one)
bool pick_defer_policy(const std::vector<Op>& ops, const std::vector<int>& core, int n, int r, bool read_map) {
  return read_map && count_stages(ops, core, n, r, true, true) < count_stages(ops, core, n, r, false, true);
}

void first_last_stage_targets(const std::vector<Op>& ops, const std::vector<int>& pass_ops, int n, int r, bool is_init,
                              std::vector<int>& first_t, std::vector<int>& last_t, bool read_map) {
  std::vector<int> lead, core, trail, rest;
  split_pass_ops(ops, pass_ops, n, is_init, lead, core, trail);
  first_t.clear(); last_t.clear();
  std::vector<int> pool = core;
  bool first = true;
  const bool defer = pick_defer_policy(ops, core, n, r, read_map);
  while (!pool.empty()) {
    StageSel sel;
    stage_select(ops, pool, n, r, sel, rest, defer, read_map);
    if (sel.count() == 0) break;
    if (first) { first_t = sel.targets; first = false; }
    last_t = sel.targets;
    pool = rest;
  }
}

struct PassInfo {
  std::vector<int> ops;      // op indices, program order
  std::vector<int> targets;  // wires that must be local
  std::vector<int> local;    // k wires
  std::vector<int> global;   // n-k wires, workgroup-index bit m <-> global[m]
  std::vector<int> lds_wire; // wire at LDS bit j
  std::vector<int> in_low;   // A_i (phys-in bits 0..lo_in-1)
  std::vector<int> out_low;  // B_i
};

}  // namespace

namespace {
struct BuildSpec {
  int n = 0;
  std::vector<Op> ops;
  std::vector<Fused> fused;
  int n_params = 0, n_gates = 0;
  bool in_state = false;   // first pass loads a canonical-order state instead of |0...0>
  bool out_state = false;  // last pass writes the canonical-order state instead of |psi|^2
  // CNOTs (control, target) that end the circuit, in program order: not executed, applied to the outcome INDEX
  // of the probabilities instead (|CX psi|^2 is a permutation of |psi|^2)
  std::vector<std::pair<int, int>> out_perm;
};
bool build_plan(const BuildSpec& spec, const PlanOptions& opt, Plan& plan, std::string& msg);
}  // namespace

bool make_plan(int ansatz, int n, int layers, const PlanOptions& opt, Plan& plan, std::string& msg) {
  if (n < 1 || n > 30) { msg = "num qubits must be in [1, 30]"; return false; }
  if (layers < 0) { msg = "layers must be >= 0"; return false; }
  std::vector<Gate> gates;
  if (!build_gate_list(ansatz, n, layers, gates)) { msg = "unknown ansatz id"; return false; }
  BuildSpec spec;
  spec.n = n;
  fuse_gates(gates, n, spec.ops, spec.fused);
  // Everything after the last one-qubit gate is CNOTs and CZs: the CZs only change phases (dropped), the CNOTs
  // permute basis states (moved into the address the probabilities are written to).
  {
    size_t last_u = spec.ops.size();
    while (last_u > 0 && spec.ops[last_u - 1].kind != K_U1) --last_u;
    for (size_t q = last_u; q < spec.ops.size(); ++q)
      if (spec.ops[q].kind == K_CX) spec.out_perm.push_back({spec.ops[q].a, spec.ops[q].b});
    spec.ops.resize(last_u);
  }
  spec.n_params = num_params(ansatz, n, layers);
  spec.n_gates = (int)gates.size();
  if (opt.kmulti == 0 && n > opt.kmax && opt.kmax >= 13) {
    // Tile size by measurement on the MI355X (tools/tile_sweep_n.py).  Round 1: 2^13 tiles (one
    // 512-thread workgroup per CU) ran a stage ~10 % slower than 2^11 tiles (four 128-thread workgroups per CU) and paid
    // only where they saved an eighth of the HBM round trips.  With the large-tile instantiation of the kernel (prefetch
    // spread over the stages, write-back folded into the last gate, matrix double-buffering) they are the fastest or
    // tied for every multi-tile state: L = 6, k = 11 / 12 / 13: n = 14: 0.58 / 0.53 / 0.51 ms, 15: 1.25 / 1.09 / 1.11,
    // 16: 2.40 / 2.39 / 2.30, 17: 5.45 / 4.95 / 4.80, 18: 11.2 / 12.0 / 11.2, 19: 29.3 / 25.3 / 23.6, 20: 78.9 / 59.5 / 55.0.
    // So: 2^13 wherever the fast kernel can run them (tables and tile within 160 KiB of LDS), else 2^11.
    PlanOptions o11 = opt, o13 = opt;
    o11.kmulti = 11;
    o13.kmulti = 13;
    // (3 register wires: 2^13 tiles, one 1024-thread workgroup per CU; the caller checks the compact tables' eligibility
    // and falls back to the 4-wire plan otherwise)
    if (opt.r == 3) return build_plan(spec, o13, plan, msg);
    Plan p13;
    std::string m13;
    if (build_plan(spec, o13, p13, m13)) {
      FastTables ft;
      if (build_fast_tables(p13, FAST_TABLE_MAX_BYTES, ft) && p13.fast_lds_bytes(ft.max_tab_rows) <= MAX_LDS_BYTES) {
        plan = std::move(p13);
        return true;
      }
    }
    return build_plan(spec, o11, plan, msg);
  }
  return build_plan(spec, opt, plan, msg);
}

// v -> (M (x) M (x) ... (x) M) v for one shared 2x2 matrix (fused gate 0): state in, state out.
// Wires are visited from the least significant physical bit upwards so that the first and the
// last pass both hold the canonical low bits locally.
bool make_kron_plan(int n, const PlanOptions& opt, Plan& plan, std::string& msg) {
  if (n < 1 || n > 30) { msg = "num bits must be in [1, 30]"; return false; }
  BuildSpec spec;
  spec.n = n;
  Fused f; f.fill(0xffffffffu); f[0] = 0; f[1] = 0;
  spec.fused.push_back(f);
  for (int w = n - 1; w >= 0; --w) spec.ops.push_back({K_U1, w, -1, 0});
  spec.in_state = spec.out_state = true;
  spec.n_gates = n;
  return build_plan(spec, opt, plan, msg);
}

namespace {
bool build_plan(const BuildSpec& spec, const PlanOptions& opt, Plan& plan, std::string& msg) {
  const int n = spec.n;
  const std::vector<Op>& ops = spec.ops;
  const std::vector<Fused>& fused = spec.fused;
  const int kmulti = opt.kmulti > 0 ? opt.kmulti : (n <= 16 ? 11 : 12);
  const int k = (n <= opt.kmax) ? n : std::min(opt.kmax, kmulti);
  const int r = std::min(opt.r, k);
  if (r > 4) { msg = "at most 4 register wires"; return false; }
  int threads = 64;
  while (threads < (1 << (k - r))) threads <<= 1;
  if (threads > opt.max_threads) { msg = "tile needs more threads than allowed"; return false; }
  if (fused.size() > 65535) { msg = "too many fused gates"; return false; }

  // ---- passes -------------------------------------------------------------------------------
  std::vector<PassInfo> passes;
  {
    std::vector<int> pool(ops.size());
    for (size_t i = 0; i < ops.size(); ++i) pool[i] = (int)i;
    std::vector<int> sel, rest, tg;
    while (!pool.empty()) {
      choose_pass(ops, pool, n, k, sel, rest, tg);
      if (sel.empty()) { msg = "planner made no progress"; return false; }
      PassInfo pi; pi.ops = sel; pi.targets = tg;
      passes.push_back(pi);
      pool = rest;
    }
    if (passes.empty()) passes.push_back(PassInfo{});
  }
  const int lo_can = std::min(opt.lo, k);
  std::vector<int> canon_low;  // wire at canonical physical bit p is n-1-p
  for (int p = 0; p < lo_can; ++p) canon_low.push_back(n - 1 - p);
  // The last pass writes the canonical order.  If its target wires plus the lo_can lowest canonical wires do not
  // fit in a tile, shorter contiguous runs (down to 2^3 elements) are accepted before a pure re-layout pass --
  // a whole extra HBM round trip of every state -- is appended.
  std::vector<int> canon_low_out = canon_low;
  {
    auto count_with = [&](int lo) {
      std::vector<char> in(n, 0);
      int cnt = 0;
      for (int w : passes.back().targets) { in[w] = 1; ++cnt; }
      for (int p = 0; p < lo; ++p) if (!in[n - 1 - p]) { in[n - 1 - p] = 1; ++cnt; }
      return cnt;
    };
    int lo_fit = -1;
    for (int lo = lo_can; lo >= std::min(3, lo_can); --lo)
      if (count_with(lo) <= k) { lo_fit = lo; break; }
    if (lo_fit < 0) passes.push_back(PassInfo{});  // pure re-layout pass
    else canon_low_out.resize(lo_fit);
  }
  if (spec.in_state) {
    std::vector<char> in(n, 0);
    int cnt = 0;
    for (int w : passes.front().targets) { in[w] = 1; ++cnt; }
    for (int w : canon_low) if (!in[w]) { in[w] = 1; ++cnt; }
    if (cnt > k) passes.insert(passes.begin(), PassInfo{});  // pure re-layout pass in front
  }
  const int np = (int)passes.size();
  // local sets: targets, then padding that favours overlap with the neighbours
  for (int i = 0; i < np; ++i) {
    PassInfo& P = passes[i];
    std::vector<char> in(n, 0);
    auto add = [&](int w) { if (!in[w] && (int)P.local.size() < k) { in[w] = 1; P.local.push_back(w); } };
    for (int w : P.targets) add(w);
    if (i == np - 1) for (int w : canon_low_out) add(w);
    if (i == 0 && spec.in_state) for (int w : canon_low) add(w);
    if (i + 1 < np) for (int w : passes[i + 1].targets) add(w);
    if (i > 0) for (int w : passes[i - 1].local) add(w);
    for (int w = n - 1; w >= 0; --w) add(w);
    for (int w = 0; w < n; ++w) if (!in[w]) P.global.push_back(w);
  }
  // Sequentially: LDS bit assignment of pass i (in_low wires first -- identity with the low
  // phys-in bits -- then the rest ascending), then out_low(i) = in_low(i+1), a subset of
  // local(i) & local(i+1) ordered for conflict-free LDS reads in the store phase.
  if (spec.in_state) passes[0].in_low = canon_low;
  for (int i = 0; i < np; ++i) {
    PassInfo& P = passes[i];
    std::vector<char> placed(n, 0);
    for (int w : P.in_low) { P.lds_wire.push_back(w); placed[w] = 1; }
    std::vector<int> others;
    for (int w : P.local) if (!placed[w]) others.push_back(w);
    std::sort(others.begin(), others.end());
    for (int w : others) P.lds_wire.push_back(w);
    if (i == np - 1) { P.out_low = canon_low_out; break; }
    std::vector<char> nxt(n, 0);
    for (int w : passes[i + 1].local) nxt[w] = 1;
    std::vector<int> ldspos(n, -1);
    for (int j = 0; j < k; ++j) ldspos[P.lds_wire[j]] = j;
    std::vector<int> pos;
    for (int w : P.local) if (nxt[w]) pos.push_back(ldspos[w]);
    // Prefer wires that are NOT targets of this pass's last stage nor of the next pass's first stage: those stages
    // can then move their amplitudes straight between registers and HBM (the low physical bits must come from the
    // thread index, not from the register wires).  At least 2^4 contiguous elements are kept (256-byte runs).
    {
      std::vector<int> f_cur, l_cur, f_nxt, l_nxt;
      first_last_stage_targets(ops, P.ops, n, r, i == 0 && !spec.in_state, f_cur, l_cur, opt.use_read_map());
      first_last_stage_targets(ops, passes[i + 1].ops, n, r, false, f_nxt, l_nxt, opt.use_read_map());
      std::vector<char> busy(n, 0);
      for (int w : l_cur) busy[w] = 1;
      for (int w : f_nxt) busy[w] = 1;
      std::vector<int> quiet;
      for (int q : pos) if (!busy[P.lds_wire[q]]) quiet.push_back(q);
      if ((int)quiet.size() >= std::min(4, std::min(opt.lo, k))) pos = quiet;
    }
    pos = order_for_banks(pos);
    const int lo = std::min((int)pos.size(), std::min(opt.lo, k));
    for (int j = 0; j < lo; ++j) P.out_low.push_back(P.lds_wire[pos[j]]);
    passes[i + 1].in_low = P.out_low;
  }
  // physical layout of the buffer written by pass i: out_low wires at bits 0.., the rest ascending;
  // the last pass writes the canonical order (wire w at bit n-1-w).
  std::vector<std::vector<int>> layout(np, std::vector<int>(n, -1));  // layout[i][wire] = phys bit
  for (int i = 0; i < np; ++i) {
    if (i == np - 1) { for (int w = 0; w < n; ++w) layout[i][w] = n - 1 - w; continue; }
    int p = 0;
    for (int w : passes[i].out_low) layout[i][w] = p++;
    if (opt.contig_out) {
      // this pass's other local wires next: the ones the next pass keeps local first (its runs get longer), then the rest --
      // a tile of this pass is then ONE contiguous block of the buffer
      std::vector<char> nxt(n, 0);
      for (int w : passes[i + 1].local) nxt[w] = 1;
      for (int w : passes[i].local) if (layout[i][w] < 0 && nxt[w]) layout[i][w] = p++;
      for (int w : passes[i].local) if (layout[i][w] < 0) layout[i][w] = p++;
    }
    for (int w = 0; w < n; ++w) if (layout[i][w] < 0) layout[i][w] = p++;
  }

  // ---- serialise ------------------------------------------------------------------------------
  std::vector<uint32_t>& W = plan.words;
  plan.max_stages = 0;
  W.assign(PH_SIZE, 0);
  W[PH_MAGIC] = PLAN_MAGIC; W[PH_N] = n; W[PH_K] = k; W[PH_NPASSES] = np;
  W[PH_NFUSED] = (uint32_t)fused.size(); W[PH_NPARAMS] = (uint32_t)spec.n_params;
  W[PH_THREADS] = threads; W[PH_R] = r; W[PH_NGATES] = (uint32_t)spec.n_gates;
  W[PH_OFF_FUSED] = (uint32_t)W.size();
  for (const Fused& f : fused) W.insert(W.end(), f.begin(), f.end());
  W[PH_OFF_PASSTAB] = (uint32_t)W.size();
  W.resize(W.size() + np, 0);
  plan.pass_off.assign(np, 0);

  for (int i = 0; i < np; ++i) {
    const PassInfo& P = passes[i];
    const uint32_t base = (uint32_t)W.size();
    plan.pass_off[i] = base;
    W[W[PH_OFF_PASSTAB] + i] = base;
    W.resize(base + PW_STAGES, 0);
    uint32_t flags = 0;
    if (i == 0 && !spec.in_state) flags |= PASS_INIT;
    if (i == np - 1) flags |= spec.out_state ? PASS_FINAL_STATE : PASS_FINAL;
    std::vector<int> ldspos(n, -1), gpos(n, -1);
    for (int j = 0; j < k; ++j) ldspos[P.lds_wire[j]] = j;
    for (int m = 0; m < n - k; ++m) gpos[P.global[m]] = m;
    auto extpos = [&](int w) { return ldspos[w] >= 0 ? ldspos[w] : k + gpos[w]; };
    const int lo_in = (int)P.in_low.size(), lo_out = (int)P.out_low.size();
    W[base + PW_FLAGS] = flags; W[base + PW_K] = k; W[base + PW_N] = n;
    W[base + PW_LO_IN] = lo_in; W[base + PW_LO_OUT] = lo_out; W[base + PW_THREADS] = threads;
    auto put_byte = [&](int table, int j, int value) { W[base + table + (j >> 2)] |= (uint32_t)value << (8 * (j & 3)); };
    if (k > 16 || n - k > 16) { msg = "tile or workgroup index wider than 16 bits"; return false; }
    for (int j = 0; j < k; ++j) {
      W[base + PW_WIRE_OF_LDS + j] = P.lds_wire[j];
      put_byte(PW_IN_PHYS, j, (i > 0) ? layout[i - 1][P.lds_wire[j]] : (n - 1 - P.lds_wire[j]));
    }
    for (int m = 0; m < n - k; ++m) {
      W[base + PW_WIRE_OF_G + m] = P.global[m];
      put_byte(PW_IN_GPHYS, m, (i > 0) ? layout[i - 1][P.global[m]] : (n - 1 - P.global[m]));
      put_byte(PW_OUT_GPHYS, m, layout[i][P.global[m]]);
    }
    // CNOTs at the head / tail of the pass that commute past everything before / after them in the pass are
    // folded into the tile load / store addressing (GF(2)-linear maps of the tile index): no LDS round trip.
    std::vector<int> lead, core, trail;
    split_pass_ops(ops, P.ops, n, (flags & PASS_INIT) != 0, lead, core, trail);
    struct LExpr { uint32_t l, g; };   // slot bit = parity(l & tile index) ^ parity(g & workgroup index)
    auto compose_lin = [&](const std::vector<int>& seq, bool reverse) {
      std::vector<LExpr> ex(k);
      for (int b2 = 0; b2 < k; ++b2) ex[b2] = LExpr{1u << b2, 0u};
      for (size_t q = 0; q < seq.size(); ++q) {
        const Op& o = ops[seq[reverse ? seq.size() - 1 - q : q]];
        const LExpr ce = ldspos[o.a] >= 0 ? ex[ldspos[o.a]] : LExpr{0u, 1u << gpos[o.a]};
        ex[ldspos[o.b]].l ^= ce.l; ex[ldspos[o.b]].g ^= ce.g;
      }
      return ex;
    };
    auto put_half = [&](int table, int j, uint32_t value) { W[base + table + (j >> 1)] |= (value & 0xffffu) << (16 * (j & 1)); };
    {
      const std::vector<LExpr> fin = compose_lin(lead, false);     // slot = pi_pre(u)
      for (int j = 0; j < k; ++j) {
        uint32_t col = 0;
        for (int b2 = 0; b2 < k; ++b2) col |= ((fin[b2].l >> j) & 1u) << b2;
        put_half(PW_IN_MASK, j, lds_swizzle(col));
      }
      for (int m = 0; m < n - k; ++m) {
        uint32_t col = 0;
        for (int b2 = 0; b2 < k; ++b2) col |= ((fin[b2].g >> m) & 1u) << b2;
        put_half(PW_IN_GMASK, m, lds_swizzle(col));
      }
    }
    {  // out enumeration: out_low wires first (phys bit j), then the other local wires by LDS position
      const std::vector<LExpr> fout = compose_lin(trail, true);    // slot = pi_post^{-1}(y)
      auto out_col = [&](int q) { uint32_t col = 0; for (int b2 = 0; b2 < k; ++b2) col |= ((fout[b2].l >> q) & 1u) << b2; return lds_swizzle(col); };
      std::vector<char> isl(n, 0);
      int j = 0;
      for (int w : P.out_low) { put_half(PW_OUT_MASK, j, out_col(ldspos[w])); put_byte(PW_OUT_PHYS, j, layout[i][w]); isl[w] = 1; ++j; }
      for (int q = 0; q < k; ++q) {
        int w = P.lds_wire[q];
        if (isl[w]) continue;
        put_half(PW_OUT_MASK, j, out_col(q)); put_byte(PW_OUT_PHYS, j, layout[i][w]); ++j;
      }
      for (int m = 0; m < n - k; ++m) {
        uint32_t col = 0;
        for (int b2 = 0; b2 < k; ++b2) col |= ((fout[b2].g >> m) & 1u) << b2;
        put_half(PW_OUT_GMASK, m, lds_swizzle(col));
      }
      // phys-out address columns: where the index bit of each wire lands.  One-hot, except for the circuit-ending
      // CNOTs of the last pass: wire w's bit is xor-ed into every wire the CNOT chain propagates it to.
      std::vector<uint32_t> wire_col(n, 0);
      for (int w = 0; w < n; ++w) wire_col[w] = 1u << layout[i][w];
      if (i == np - 1 && !spec.out_perm.empty()) {
        std::vector<uint32_t> reach(n);          // reach[w] = set of wires whose final bit contains bit w
        std::vector<uint32_t> expr(n);           // expr[w] = set of original wire bits xor-ed into wire w
        for (int w = 0; w < n; ++w) expr[w] = 1u << w;
        for (const auto& cx : spec.out_perm) expr[cx.second] ^= expr[cx.first];
        for (int w = 0; w < n; ++w) {
          uint32_t col = 0;
          for (int w2 = 0; w2 < n; ++w2) if ((expr[w2] >> w) & 1u) col ^= 1u << layout[i][w2];
          wire_col[w] = col;
        }
      }
      {
        std::vector<char> isl2(n, 0);
        int j2 = 0;
        for (int w : P.out_low) { W[base + PW_OUT_COL + j2] = wire_col[w]; isl2[w] = 1; ++j2; }
        for (int q = 0; q < k; ++q) {
          const int w = P.lds_wire[q];
          if (isl2[w]) continue;
          W[base + PW_OUT_COL + j2] = wire_col[w]; ++j2;
        }
        for (int m = 0; m < n - k; ++m) W[base + PW_OUT_GCOL + m] = wire_col[P.global[m]];
      }
    }
    // ---- stages ----
    // Fixed form per stage (no runtime dispatch in the kernel):
    //   load (CNOTs of phase 0 folded into the LDS read address) -> sign (CZs of phase 1)
    //   -> at most one fused U per register wire (phase 2)
    //   -> CNOTs of phase 3 folded into the LDS write address, sign of phase 4 evaluated on the
    //      permuted index.  A CNOT whose target is a register wire only permutes amplitudes inside
    //      the 2^r-element group a thread owns, so the in-place write-back needs no extra barrier.
    std::vector<int> pool = core, rest;
    uint32_t nstages = 0;
    const bool defer_cx3 = pick_defer_policy(ops, core, n, r, opt.use_read_map());
    while (!pool.empty()) {
      StageSel sel;
      stage_select(ops, pool, n, r, sel, rest, defer_cx3, opt.use_read_map());
      if (sel.count() == 0) { msg = "stage planner made no progress"; return false; }
      if (getenv("BORNVI_PLAN_DEBUG")) {
        auto show = [&](const char* nm, const std::vector<int>& v) {
          fprintf(stderr, " %s[", nm);
          for (int idx : v) fprintf(stderr, "%s%d%s ", ops[idx].kind == K_U1 ? "U" : (ops[idx].kind == K_CX ? "X" : "Z"), ops[idx].a,
                                    ops[idx].b >= 0 ? (std::string(">") + std::to_string(ops[idx].b)).c_str() : "");
          fprintf(stderr, "]");
        };
        fprintf(stderr, "pass %d stage %u:", i, nstages);
        show("cx0", sel.pre_cx); show("cz1", sel.pre_cz); show("u", sel.us); show("cx3", sel.post_cx); show("cz4", sel.post_cz);
        fprintf(stderr, "  rest %zu\n", rest.size());
      }
      // register wires: targets, padded with local wires (highest LDS bits first)
      std::vector<char> isr(n, 0);
      std::vector<int> regw;   // wires carrying a fused U first (the fast kernel's stage kinds count them)
      {
        std::vector<char> has_u(n, 0);
        for (int idx : sel.us) has_u[ops[idx].a] = 1;
        for (int w : sel.targets) if (has_u[w]) regw.push_back(w);
        for (int w : sel.targets) if (!has_u[w]) regw.push_back(w);
      }
      for (int w : regw) isr[w] = 1;
      {   // padding: highest LDS bits first, wires that hold low physical bits of the buffers only as a last resort
        std::vector<char> lowwire(n, 0);
        for (int w : P.in_low) lowwire[w] = 1;
        for (int w : P.out_low) lowwire[w] = 1;
        for (int avoid = 1; avoid >= 0; --avoid)
          for (int j = k - 1; j >= 0 && (int)regw.size() < r; --j)
            if (!isr[P.lds_wire[j]] && !(avoid && lowwire[P.lds_wire[j]])) { isr[P.lds_wire[j]] = 1; regw.push_back(P.lds_wire[j]); }
      }
      std::vector<int> regbit(n, -1);
      uint32_t rho = 0;
      for (int b = 0; b < r; ++b) { regbit[regw[b]] = b; rho |= (uint32_t)ldspos[regw[b]] << (8 * b); }
      std::vector<int> freepos;
      for (int j = 0; j < k; ++j) if (!isr[P.lds_wire[j]]) freepos.push_back(j);
      freepos = order_for_banks(freepos);
      // First / last stage of a pass: the fast kernel moves their amplitudes straight between HBM and registers
      // (no tile fill before the first stage, no tile drain after the last), which coalesces when the low thread
      // bits sit on the LDS positions of the wires that hold the low physical bits of the buffer read (in_low,
      // positions 0 .. lo_in-1) resp. written (out_low).  Only the thread-bit ORDER changes; it keeps the
      // 8-lane ds_write groups and 16-lane ds_read groups conflict-free (positions 0,1,2 | 3 | 4: residues mod 4).
      uint32_t io_flags = 0;
      {
        const bool first = (nstages == 0) && !(flags & PASS_INIT);
        const bool last = rest.empty();
        auto lead_with = [&](const std::vector<int>& lowpos) {
          for (int q : lowpos) if (isr[P.lds_wire[q]]) return false;    // a low wire is a register wire: not possible
          std::vector<char> taken(k, 0);
          std::vector<int> others;
          for (int q : lowpos) taken[q] = 1;
          for (int q : freepos) if (!taken[q]) others.push_back(q);
          std::vector<int> ord = lowpos;
          ord.insert(ord.end(), others.begin(), others.end());
          freepos = ord;
          return true;
        };
        if (first && !P.in_low.empty()) {
          std::vector<int> lowpos;
          for (size_t q = 0; q < P.in_low.size(); ++q) lowpos.push_back((int)q);    // in_low wires sit at LDS bits 0..
          if ((int)lowpos.size() <= k - r && lead_with(lowpos)) io_flags |= STAGE_FROM_HBM;
        }
        if (last && !(io_flags & STAGE_FROM_HBM) && !P.out_low.empty()) {
          std::vector<int> lowpos;
          for (int w : P.out_low) lowpos.push_back(ldspos[w]);
          if ((int)lowpos.size() <= k - r && lead_with(lowpos)) io_flags |= STAGE_TO_HBM;
        }
      }
      uint32_t tpos[4] = {0, 0, 0, 0};
      for (size_t j = 0; j < freepos.size() && j < 16; ++j) tpos[j / 4] |= (uint32_t)freepos[j] << (8 * (j % 4));

      // symbolic composition of the CNOTs: register bit t = parity(areg & slot) ^ parity(bext & e)
      struct Expr { uint32_t areg, bext; };
      auto compose = [&](const std::vector<int>& seq, bool reverse, Expr (&ex)[4]) {
        for (int t = 0; t < 4; ++t) ex[t] = Expr{1u << t, 0u};
        for (size_t q = 0; q < seq.size(); ++q) {
          const Op& o = ops[seq[reverse ? seq.size() - 1 - q : q]];
          const int rt = regbit[o.b];
          Expr ce = regbit[o.a] >= 0 ? ex[regbit[o.a]] : Expr{0u, 1u << extpos(o.a)};
          ex[rt].areg ^= ce.areg; ex[rt].bext ^= ce.bext;
        }
      };
      Expr pre[4], post[4];
      compose(sel.post_cx, false, post);  // forward map: where the amplitude of slot j is written to
      // READ map (phase-0 CNOTs, inverse = the same CNOTs in reverse order) over ALL k LDS positions: source bit of
      // position p = parity(areg & slot) ^ parity(bext & e).  Register positions start as (1 << t, 0), thread-held
      // positions as (0, 1 << p), workgroup bits never change.
      std::vector<Expr> pre_pos((size_t)k);
      for (int p2 = 0; p2 < k; ++p2) pre_pos[(size_t)p2] = Expr{0u, 1u << p2};
      for (int t = 0; t < r; ++t) pre_pos[(size_t)ldspos[regw[t]]] = Expr{1u << t, 0u};
      for (size_t q = 0; q < sel.pre_cx.size(); ++q) {
        const Op& o = ops[sel.pre_cx[sel.pre_cx.size() - 1 - q]];
        const Expr ce = ldspos[o.a] >= 0 ? pre_pos[(size_t)ldspos[o.a]] : Expr{0u, 1u << extpos(o.a)};
        Expr& te = pre_pos[(size_t)ldspos[o.b]];
        te.areg ^= ce.areg; te.bext ^= ce.bext;
      }
      for (int t = 0; t < 4; ++t) pre[t] = t < r ? pre_pos[(size_t)ldspos[regw[t]]] : Expr{1u << t, 0u};
      auto slot_offset = [&](const Expr (&ex)[4], int j) {
        uint32_t off = 0;
        for (int t = 0; t < r; ++t)
          if (__builtin_popcount(ex[t].areg & (uint32_t)j) & 1) off |= 1u << ldspos[regw[t]];
        return off;
      };
      auto read_slot_offset = [&](int j) {      // all positions: a read may come from another thread's group
        uint32_t off = 0;
        for (int p2 = 0; p2 < k; ++p2)
          if (__builtin_popcount(pre_pos[(size_t)p2].areg & (uint32_t)j) & 1) off |= 1u << p2;
        return off;
      };
      auto emit_signq = [&](const std::vector<int>& czs, const Expr (&ex)[4]) {
        uint32_t U[32]; std::memset(U, 0, sizeof(U));
        uint32_t Sym[32]; std::memset(Sym, 0, sizeof(Sym));
        for (int idx : czs) {
          int ea = extpos(ops[idx].a), eb = extpos(ops[idx].b);
          int lo2 = std::min(ea, eb), hi2 = std::max(ea, eb);
          U[lo2] ^= 1u << hi2;               // CZ twice = identity, hence xor
          Sym[lo2] ^= 1u << hi2; Sym[hi2] ^= 1u << lo2;
        }
        for (int a = 0; a < 32; ++a) W.push_back(U[a]);
        uint32_t qbits = 0;
        for (int j = 0; j < 16; ++j) {
          uint32_t off = (j < (1 << r)) ? slot_offset(ex, j) : 0u, m = 0, qv = 0;
          for (int a = 0; a < 32; ++a)
            if (off >> a & 1) { m ^= Sym[a]; qv ^= (uint32_t)__builtin_popcount(off & U[a]) & 1u; }
          W.push_back(m);
          qbits |= qv << j;
        }
        W.push_back(qbits);
      };

      const uint32_t sbase = (uint32_t)W.size();
      W.resize(sbase + STAGE_HDR_WORDS, 0);
      W[sbase + 1] = rho;
      for (int q = 0; q < 4; ++q) W[sbase + 2 + q] = tpos[q];
      uint32_t fidx[4] = {0xffffu, 0xffffu, 0xffffu, 0xffffu};
      for (int idx : sel.us) fidx[regbit[ops[idx].a]] = (uint32_t)ops[idx].idx;
      W[sbase + 6] = fidx[0] | (fidx[1] << 16);
      W[sbase + 7] = fidx[2] | (fidx[3] << 16);
      if (nstages >= (uint32_t)MAX_STAGES) { msg = "too many stages in one pass"; return false; }
      W[base + PW_MATS + 2 * nstages] = W[sbase + 6];
      W[base + PW_MATS + 2 * nstages + 1] = W[sbase + 7];
      for (int t = 0; t < 4; ++t) { W[sbase + 8 + t] = t < r ? pre[t].bext : 0u; W[sbase + 12 + t] = t < r ? post[t].bext : 0u; }
      for (int j = 0; j < 16; ++j) {
        W[sbase + 16 + j] = (j < (1 << r)) ? lds_swizzle(read_slot_offset(j)) : 0u;
        W[sbase + 32 + j] = (j < (1 << r)) ? lds_swizzle(slot_offset(post, j)) : 0u;
      }
      bool cross_read = false;
      for (int p2 = 0; p2 < k; ++p2)          // thread-held positions moved by the read map
        if (!isr[P.lds_wire[p2]]) {
          W[sbase + 48 + p2] = pre_pos[(size_t)p2].bext ^ (1u << p2);
          if (W[sbase + 48 + p2] || pre_pos[(size_t)p2].areg) cross_read = true;
        }
      uint32_t flags = io_flags | (cross_read ? STAGE_CROSS_READ : 0u);
      Expr ident[4];
      for (int t = 0; t < 4; ++t) ident[t] = Expr{1u << t, 0u};
      if (!sel.pre_cz.empty()) { flags |= STAGE_SIGN_PRE; emit_signq(sel.pre_cz, ident); }
      if (!sel.post_cz.empty()) { flags |= STAGE_SIGN_POST; emit_signq(sel.post_cz, post); }
      for (uint32_t tt = 0; tt < (1u << (k - r)); ++tt) {   // per-thread base table
        uint32_t bse = 0;
        for (size_t j = 0; j < freepos.size(); ++j) bse |= ((tt >> j) & 1u) << freepos[j];
        W.push_back(lds_swizzle(bse) | (bse << 16));
      }
      const uint32_t nwords = (uint32_t)W.size() - sbase;
      if (nwords >= (1u << 16)) { msg = "stage too large"; return false; }
      W[sbase] = (uint32_t)r | (flags << 8) | (nwords << 16);
      ++nstages;
      pool = rest;
    }
    W[base + PW_NSTAGES] = nstages;
    if ((int)nstages > plan.max_stages) plan.max_stages = (int)nstages;
  }
  W[PH_TOTAL] = (uint32_t)W.size();
  // first pass whose matrices depend on parameter p (a fused gate sits in exactly one stage of one pass)
  plan.param_first_pass.assign(spec.n_params, np > 0 ? np - 1 : 0);
  std::vector<char> param_seen(spec.n_params, 0);
  for (int i = 0; i < np; ++i) {
    const uint32_t* PWp = W.data() + plan.pass_off[i];
    for (uint32_t sm = 0; sm < PWp[PW_NSTAGES] * 4u; ++sm) {
      const uint32_t w = PWp[PW_MATS + (sm >> 1)];
      const uint32_t f = (sm & 1u) ? (w >> 16) : (w & 0xffffu);
      if (f == 0xffffu) continue;
      const uint32_t* fw = W.data() + W[PH_OFF_FUSED] + f * FUSED_WORDS;
      for (uint32_t e = 0; e < fw[1]; ++e) {
        const uint32_t par = fw[3 + 2 * e];
        if (par != 0xffffffffu && par < (uint32_t)spec.n_params && !param_seen[par]) { param_seen[par] = 1; plan.param_first_pass[par] = i; }
      }
    }
  }
  for (int q = 0; q < spec.n_params; ++q) if (!param_seen[q]) plan.param_first_pass[q] = 0;   // (not in any gate: no saving claimed)
  plan.n = n; plan.k = k; plan.r = r; plan.n_passes = np; plan.n_fused = (int)fused.size();
  plan.n_params = spec.n_params; plan.threads = threads; plan.n_gates = spec.n_gates;
  return true;
}
}  // namespace

// ---- fast-path tables ---------------------------------------------------------------------------
// Evaluates, per (stage, tile, thread), exactly what circuit_pass_kernel computes at run time from the
// stage header (see also tests/plan_emulator.py, which interprets the same words).
bool build_fast_tables(const Plan& plan, size_t max_bytes, FastTables& out) {
  out.words.clear(); out.pass_off.clear(); out.any_sign = false; out.max_tab_rows = 1;
  const int n = plan.n, k = plan.k;
  const int r = plan.r;
  if ((r != 4 && r != 3) || k < (r == 4 ? 10 : 9) || plan.threads != (1 << (k - r)) || n - k > 16) return false;
  out.r = r;
  const int kt = k - r;
  const int nslots = 1 << r;
  const size_t per_stage = (size_t)1 << (n - r);
  size_t total = 0;
  for (int i = 0; i < plan.n_passes; ++i) {
    const uint32_t* P = plan.words.data() + plan.pass_off[i];
    const uint32_t nst = P[PW_NSTAGES];
    if (nst > (uint32_t)MAX_STAGES || nst * 4u * (uint32_t)r > (uint32_t)plan.threads) return false;   // one matrix piece per thread
    bool any_sign = false;
    const uint32_t* S = P + PW_STAGES;
    for (uint32_t s = 0; s < nst; ++s) { if ((S[0] >> 8) & (STAGE_SIGN_PRE | STAGE_SIGN_POST)) any_sign = true; S += S[0] >> 16; }
    total += FH_WORDS + (size_t)nst * FS_WORDS + (size_t)nst * per_stage * (any_sign ? 2 : 1) + 2 * per_stage;
  }
  if (total * sizeof(uint32_t) > max_bytes || total >= (size_t)0xffffffffu) return false;
  std::vector<uint32_t>& W = out.words;
  W.reserve(total);
  auto parity = [](uint32_t v) { return (uint32_t)__builtin_popcount(v) & 1u; };
  for (int i = 0; i < plan.n_passes; ++i) {
    const uint32_t* P = plan.words.data() + plan.pass_off[i];
    const uint32_t nst = P[PW_NSTAGES];
    const uint32_t hbase = (uint32_t)W.size();
    out.pass_off.push_back(hbase);
    W.resize(hbase + FH_WORDS + (size_t)nst * FS_WORDS, 0);
    uint32_t sign_pre = 0, sign_post = 0;
    {
      const uint32_t* S = P + PW_STAGES;
      for (uint32_t s = 0; s < nst; ++s) {
        const uint32_t sf = (S[0] >> 8) & 0xffu;
        if (sf & STAGE_SIGN_PRE) sign_pre |= 1u << s;
        if (sf & STAGE_SIGN_POST) sign_post |= 1u << s;
        S += S[0] >> 16;
      }
    }
    const bool any_sign = (sign_pre | sign_post) != 0;
    if (any_sign) out.any_sign = true;
    out.max_tab_rows = std::max(out.max_tab_rows, (int)nst + __builtin_popcount(sign_pre | sign_post));
    const uint32_t rw_base = (uint32_t)W.size();
    W.resize(W.size() + (size_t)nst * per_stage, 0);
    const uint32_t sg_base = any_sign ? (uint32_t)W.size() : 0u;
    if (any_sign) W.resize(W.size() + (size_t)nst * per_stage, 0);
    W[hbase + FH_NSTAGES] = nst; W[hbase + FH_RW_BASE] = rw_base; W[hbase + FH_SG_BASE] = sg_base;
    W[hbase + FH_SIGN_PRE] = sign_pre; W[hbase + FH_SIGN_POST] = sign_post;
    const uint32_t* S = P + PW_STAGES;
    for (uint32_t s = 0; s < nst; ++s) {
      const uint32_t hdr = S[0];
      if ((hdr & 0xffu) != (uint32_t)r) { out.words.clear(); out.pass_off.clear(); return false; }
      const uint32_t sflags = (hdr >> 8) & 0xffu, nwords = hdr >> 16, rho = S[1];
      uint32_t* FS = W.data() + hbase + FH_WORDS + (size_t)s * FS_WORDS;
      FS[FS_FI01] = S[6]; FS[FS_FI23] = S[7];
      {   // gates must occupy register bits 0 .. ng-1 (stage planner: U-carrying wires first)
        const uint32_t fi[4] = {S[6] & 0xffffu, S[6] >> 16, S[7] & 0xffffu, S[7] >> 16};
        uint32_t ng = 0;
        while (ng < (uint32_t)r && fi[ng] != 0xffffu) ++ng;
        for (uint32_t b = ng; b < 4; ++b)
          if (fi[b] != 0xffffu) { out.words.clear(); out.pass_off.clear(); return false; }
        FS[FS_KIND] = ng | ((sflags & STAGE_SIGN_PRE) ? 8u : 0u) | ((sflags & STAGE_SIGN_POST) ? 16u : 0u);
        FS[FS_CROSS] = (sflags & STAGE_CROSS_READ) ? 1u : 0u;
        if (!fast_stage_kind_supported(FS[FS_KIND])) { out.words.clear(); out.pass_off.clear(); return false; }
      }
      for (int b = 0; b < r; ++b) { FS[FS_RB + b] = S[16 + (1 << b)] << 4; FS[FS_WB + b] = S[32 + (1 << b)] << 4; }
      for (int j = 0; j < nslots; ++j) {   // the slot offsets must be linear in the slot number
        uint32_t lr = 0, lw = 0;
        for (int b = 0; b < r; ++b) if (j >> b & 1) { lr ^= S[16 + (1 << b)]; lw ^= S[32 + (1 << b)]; }
        if (lr != S[16 + j] || lw != S[32 + j]) { out.words.clear(); out.pass_off.clear(); return false; }
      }
      uint32_t sri[4] = {0, 0, 0, 0}, rpos[4] = {0, 0, 0, 0};
      for (int b = 0; b < r; ++b) { rpos[b] = (rho >> (8 * b)) & 0xffu; sri[b] = lds_swizzle(1u << rpos[b]); }
      const uint32_t* Qpre = (sflags & STAGE_SIGN_PRE) ? S + STAGE_HDR_WORDS : nullptr;
      const uint32_t* Qpost = (sflags & STAGE_SIGN_POST) ? S + STAGE_HDR_WORDS + (Qpre ? SIGNQ_WORDS : 0) : nullptr;
      const uint32_t* tab = S + STAGE_HDR_WORDS + (Qpre ? SIGNQ_WORDS : 0) + (Qpost ? SIGNQ_WORDS : 0);
      auto sign_bits = [&](const uint32_t* Q, uint32_t e) {
        uint32_t acc = 0;
        for (int q = 0; q < n; ++q) acc ^= ((e >> q) & 1u) & parity(e & Q[q]);
        uint32_t m = 0;
        for (int j = 0; j < 16; ++j) m |= (((Q[48] >> j) ^ acc ^ parity(e & Q[32 + j])) & 1u) << j;
        return m;
      };
      uint32_t* RW = W.data() + rw_base + (size_t)s * per_stage;
      uint32_t* SG = any_sign ? W.data() + sg_base + (size_t)s * per_stage : nullptr;
      for (uint32_t g = 0; g < (1u << (n - k)); ++g)
        for (uint32_t t = 0; t < (1u << kt); ++t) {
          const uint32_t pb = tab[t] & 0xffffu, base = tab[t] >> 16;
          const uint32_t e = base | (g << k);
          uint32_t lflip = 0, sflip = 0, e2 = e;
          for (int b = 0; b < r; ++b) {
            if (parity(e & S[8 + b])) lflip ^= sri[b];
            if (parity(e & S[12 + b])) { sflip ^= sri[b]; e2 |= 1u << rpos[b]; }
          }
          for (int p2 = 0; p2 < k; ++p2)
            if (S[48 + p2] && parity(e & S[48 + p2])) lflip ^= lds_swizzle(1u << p2);
          RW[((size_t)g << kt) + t] = (pb ^ lflip) | ((pb ^ sflip) << 16);
          if (SG) SG[((size_t)g << kt) + t] = (Qpre ? sign_bits(Qpre, e) : 0u) | ((Qpost ? sign_bits(Qpost, e2) : 0u) << 16);
        }
      S += nwords;
    }
    // ---- direct HBM <-> register tables for the first / last stage --------------------------------------------
    const uint32_t pflags = P[PW_FLAGS];
    const uint32_t kt2 = (uint32_t)kt;
    auto tbyte = [&](int table, int j) { return (P[table + (j >> 2)] >> (8 * (j & 3))) & 0xffu; };
    auto thalf = [&](int table, int j) { return (P[table + (j >> 1)] >> (16 * (j & 1))) & 0xffffu; };
    const uint32_t ksize = 1u << k;
    const uint32_t ngl = 1u << (n - k);
    if (nst > 0) {
      const uint32_t* S0 = P + PW_STAGES;
      const uint32_t* SL = S0;
      for (uint32_t s2 = 0; s2 + 1 < nst; ++s2) SL += SL[0] >> 16;
      const uint32_t f0 = (S0[0] >> 8) & 0xffu, fl = (SL[0] >> 8) & 0xffu;
      std::vector<uint32_t> inv(ksize);
      if ((f0 & STAGE_FROM_HBM) && !(pflags & PASS_INIT)) {
        // slot(u) = xor_j bit_j(u) IN_MASK[j] ^ xor_m bit_m(g) IN_GMASK[m];  phys(u) = deposit(u, IN_PHYS) | deposit(g, IN_GPHYS)
        auto slot_lin = [&](uint32_t u) { uint32_t o = 0; for (int j = 0; j < k; ++j) if (u >> j & 1) o ^= thalf(PW_IN_MASK, j); return o; };
        auto phys_u = [&](uint32_t u) { uint32_t o = 0; for (int j = 0; j < k; ++j) if (u >> j & 1) o |= 1u << tbyte(PW_IN_PHYS, j); return o; };
        for (uint32_t u = 0; u < ksize; ++u) inv[slot_lin(u)] = u;
        const uint32_t tab = (uint32_t)W.size();
        W.resize(W.size() + per_stage, 0);
        const uint32_t* RW0 = W.data() + rw_base;
        bool ok = true;
        uint32_t basis[4] = {0, 0, 0, 0};
        for (int b = 0; b < r; ++b) basis[b] = phys_u(inv[W[hbase + FH_WORDS + FS_RB + b] >> 4]) << 4;
        for (uint32_t g = 0; g < ngl && ok; ++g) {
          uint32_t gslot = 0, gphys = 0;
          for (int m = 0; m < n - k; ++m) if (g >> m & 1) { gslot ^= thalf(PW_IN_GMASK, m); gphys |= 1u << tbyte(PW_IN_GPHYS, m); }
          for (uint32_t t = 0; t < (1u << kt2); ++t) {
            const uint32_t rslot = RW0[((size_t)g << kt2) + t] & 0xffffu;
            const uint32_t off = (phys_u(inv[rslot ^ gslot]) | gphys) << 4;
            W[tab + ((size_t)g << kt2) + t] = off;
          }
          // every group of 2^lo_in lanes must load one aligned run of 2^lo_in elements (1 KiB at lo_in = 6)
          const uint32_t lanes = 1u << std::min<uint32_t>(P[PW_LO_IN], 6u);
          const uint32_t runmask = lanes * 16u - 1u;
          if (P[PW_LO_IN] < 4u) ok = false;
          for (uint32_t t0 = 0; t0 < (1u << kt2) && ok; t0 += lanes) {
            const uint32_t base_run = W[tab + ((size_t)g << kt2) + t0] & ~runmask;
            uint64_t seen = 0;
            for (uint32_t l = 0; l < lanes; ++l) {
              const uint32_t o = W[tab + ((size_t)g << kt2) + t0 + l];
              if ((o & ~runmask) != base_run) ok = false;
              seen |= 1ull << ((o >> 4) & (lanes - 1u));
            }
            if (seen != (lanes == 64 ? ~0ull : ((1ull << lanes) - 1ull))) ok = false;
          }
        }
        for (int b = 0; b < r; ++b) if (basis[b] & ((16u << std::min<uint32_t>(P[PW_LO_IN], 6u)) - 1u)) ok = false;   // slot offsets must not move inside the run
        if (ok) {
          W[hbase + FH_IN_TAB] = tab;
          for (int b = 0; b < r; ++b) W[hbase + FH_IN_BASIS + b] = basis[b];
          // Support of |0..0>.  After an INIT pass 0 the state is zero wherever one of pass 0's tile-index wires is 1
          // (its gates act inside a tile).  If this is pass 1 and those wires' address bits reach the direct first
          // stage only through the SLOT offsets (never through a thread's base), slot j of every thread is known to be
          // zero whenever its offset carries one of them: the kernel does not load it (FH_ZINFO of this pass = mask of
          // such slots), and pass 0 need not write the tiles that only such slots would read (FH_ZINFO of pass 0 = mask
          // over its tile-index bits: tile g is skipped when g & mask != 0).  Zero-wire bits that sit in thread bases
          // stay out of both masks (those loads happen and read the zeros pass 0 writes).
          if (i == 1 && (plan.words[plan.pass_off[0] + PW_FLAGS] & PASS_INIT) && plan.n_passes > 2) {
            const uint32_t* P0 = plan.words.data() + plan.pass_off[0];
            uint32_t zbits = 0;                                   // byte-offset bits of pass 0's tile-index wires
            for (int m = 0; m < n - k; ++m) zbits |= 16u << ((P0[PW_OUT_GPHYS + (m >> 2)] >> (8 * (m & 3))) & 0xffu);
            uint32_t in_base = 0, in_slot = 0;
            for (uint32_t g = 0; g < ngl; ++g)
              for (uint32_t t = 0; t < (1u << kt2); ++t) in_base |= W[tab + ((size_t)g << kt2) + t] & zbits;
            for (int b = 0; b < r; ++b) in_slot |= basis[b] & zbits;
            const uint32_t zs = in_slot & ~in_base;               // zero-wire bits that only slots carry
            if (zs) {
              uint32_t zslots = 0;
              for (int j = 0; j < nslots; ++j) {
                uint32_t off = 0;
                for (int b = 0; b < r; ++b) if (j >> b & 1) off ^= basis[b];
                if (off & zs) zslots |= 1u << j;
              }
              uint32_t gmask0 = 0;
              for (int m = 0; m < n - k; ++m)
                if (zs & (16u << ((P0[PW_OUT_GPHYS + (m >> 2)] >> (8 * (m & 3))) & 0xffu))) gmask0 |= 1u << m;
              W[hbase + FH_ZINFO] = zslots;
              W[out.pass_off[0] + FH_ZINFO] = gmask0;
            }
          }
        } else {
          W.resize(tab);
        }
      }
      if (fl & STAGE_TO_HBM) {
        // out enumeration v: slot(v) = xor_j bit_j(v) OUT_MASK[j] ^ xor_m bit_m(g) OUT_GMASK[m];
        // phys(v) = xor_j bit_j(v) OUT_COL[j] ^ xor_m bit_m(g) OUT_GCOL[m]
        const int sh = (pflags & PASS_FINAL) ? 3 : 4;
        auto slot_lin = [&](uint32_t v) { uint32_t o = 0; for (int j = 0; j < k; ++j) if (v >> j & 1) o ^= thalf(PW_OUT_MASK, j); return o; };
        auto phys_v = [&](uint32_t v) { uint32_t o = 0; for (int j = 0; j < k; ++j) if (v >> j & 1) o ^= P[PW_OUT_COL + j]; return o; };
        for (uint32_t v = 0; v < ksize; ++v) inv[slot_lin(v)] = v;
        const uint32_t tab = (uint32_t)W.size();
        W.resize(W.size() + per_stage, 0);
        const uint32_t* RWL = W.data() + rw_base + (size_t)(nst - 1) * per_stage;
        bool ok = (n + sh) <= 32;
        uint32_t basis[4] = {0, 0, 0, 0};
        for (int b = 0; b < r; ++b) basis[b] = phys_v(inv[W[hbase + FH_WORDS + (size_t)(nst - 1) * FS_WORDS + FS_WB + b] >> 4]) << sh;
        const uint32_t lanes_o = 1u << std::min<uint32_t>(P[PW_LO_OUT], 6u);
        const uint32_t runmask = (lanes_o << sh) - 1u;
        if (P[PW_LO_OUT] < 4u && !(pflags & PASS_FINAL)) ok = false;
        if ((pflags & PASS_FINAL) && P[PW_LO_OUT] < 3u) ok = false;
        for (uint32_t g = 0; g < ngl && ok; ++g) {
          uint32_t gslot = 0, gphys = 0;
          for (int m = 0; m < n - k; ++m) if (g >> m & 1) { gslot ^= thalf(PW_OUT_GMASK, m); gphys ^= P[PW_OUT_GCOL + m]; }
          for (uint32_t t = 0; t < (1u << kt2); ++t) {
            const uint32_t wslot = RWL[((size_t)g << kt2) + t] >> 16;
            W[tab + ((size_t)g << kt2) + t] = (phys_v(inv[wslot ^ gslot]) ^ gphys) << sh;
          }
          // every 64-lane store of a slot covers one aligned run of 64 elements -- or two half-filled ones in the
          // last pass (the wrap-around CNOT of the circuit-ending ring puts the lanes' parity into the top bit)
          for (uint32_t t0 = 0; t0 < (1u << kt2) && ok; t0 += lanes_o) {
            uint32_t runs[2] = {0, 0};
            int nruns = 0;
            uint64_t seen = 0;
            for (uint32_t l = 0; l < lanes_o; ++l) {
              const uint32_t o = W[tab + ((size_t)g << kt2) + t0 + l];
              const uint32_t rb = o & ~runmask;
              int f = -1;
              for (int q = 0; q < nruns; ++q) if (runs[q] == rb) f = q;
              if (f < 0) { if (nruns == 2) { ok = false; break; } runs[nruns++] = rb; }
              seen |= 1ull << ((o >> sh) & (lanes_o - 1u));
            }
            if (nruns > ((pflags & PASS_FINAL) ? 2 : 1)) ok = false;
            if (seen != (lanes_o == 64 ? ~0ull : ((1ull << lanes_o) - 1ull))) ok = false;
          }
        }
        if (ok) {
          W[hbase + FH_OUT_TAB] = tab;
          for (int b = 0; b < r; ++b) W[hbase + FH_OUT_BASIS + b] = basis[b];
        } else {
          W.resize(tab);
        }
      }
    }
  }
  return true;
}


// ---- compact tables (plan.hpp: CompactTables; kernels_circuit8.hip) --------------------------------------------------
// Point evaluation of the same quantities build_fast_tables tabulates, then their lane / wave / tile-row decomposition,
// checked word by word against the point evaluation.
namespace {
inline uint32_t par32(uint32_t v) { return (uint32_t)__builtin_popcount(v) & 1u; }

struct StageEval {
  const uint32_t* S = nullptr;      // stage header (plan words)
  const uint32_t* tab = nullptr;    // per-thread base table: lds_swizzle(base_t) | base_t << 16
  const uint32_t* Qpre = nullptr;
  const uint32_t* Qpost = nullptr;
  uint32_t sri[3] = {0, 0, 0}, rpos[3] = {0, 0, 0};
  int k = 0, n = 0;
  // read slot | write slot << 16 of thread t in tile row g; e / e2: the extended indices the signs are evaluated on
  uint32_t rw(uint32_t g, uint32_t t, uint32_t* e_out = nullptr, uint32_t* e2_out = nullptr) const {
    const uint32_t pb = tab[t] & 0xffffu, base = tab[t] >> 16;
    const uint32_t e = base | (g << k);
    uint32_t lflip = 0, sflip = 0, e2 = e;
    for (int b = 0; b < 3; ++b) {
      if (par32(e & S[8 + b])) lflip ^= sri[b];
      if (par32(e & S[12 + b])) { sflip ^= sri[b]; e2 |= 1u << rpos[b]; }
    }
    for (int p2 = 0; p2 < k; ++p2)
      if (S[48 + p2] && par32(e & S[48 + p2])) lflip ^= lds_swizzle(1u << p2);
    if (e_out) *e_out = e;
    if (e2_out) *e2_out = e2;
    return (pb ^ lflip) | ((pb ^ sflip) << 16);
  }
  uint32_t sign_bits(const uint32_t* Q, uint32_t e) const {
    uint32_t acc = 0;
    for (int q = 0; q < n; ++q) acc ^= ((e >> q) & 1u) & par32(e & Q[q]);
    uint32_t m = 0;
    for (int j = 0; j < 16; ++j) m |= (((Q[48] >> j) ^ acc ^ par32(e & Q[32 + j])) & 1u) << j;
    return m;
  }
  uint32_t sg(uint32_t g, uint32_t t) const {
    uint32_t e, e2;
    (void)rw(g, t, &e, &e2);
    return (Qpre ? sign_bits(Qpre, e) : 0u) | ((Qpost ? sign_bits(Qpost, e2) : 0u) << 16);
  }
};
}  // namespace

bool build_compact_tables(const Plan& plan, CompactTables& out, std::string& msg) {
  out = CompactTables();
  const int n = plan.n, k = plan.k, r = plan.r;
  if (r != 3) { msg = "compact tables: the plan must have 3 register wires"; return false; }
  // (tiles of 2^6 .. 2^8 amplitudes: one wave whose first 2^(k-3) lanes hold amplitudes -- the launch-bound sizes n <= 8)
  if (k < 6 || plan.threads != std::max(64, 1 << (k - 3)) || n - k > 16 || n > 27) { msg = "compact tables: tile below 2^6 or state too large"; return false; }
  const int kt = k - 3, gbits = n - k;
  const uint32_t T = 1u << kt, NW = std::max(T / 64u, 1u), ngl = 1u << gbits, ksize = 1u << k;
  std::vector<uint32_t>& W = out.words;
  auto fail = [&](const std::string& m) { msg = "compact tables: " + m; out = CompactTables(); return false; };
  for (int i = 0; i < plan.n_passes; ++i) {
    const uint32_t* P = plan.words.data() + plan.pass_off[i];
    const uint32_t nst = P[PW_NSTAGES];
    if (nst > (uint32_t)MAX_STAGES) return fail("too many stages");
    auto tbyte = [&](int table, int j) { return (P[table + (j >> 2)] >> (8 * (j & 3))) & 0xffu; };
    auto thalf = [&](int table, int j) { return (P[table + (j >> 1)] >> (16 * (j & 1))) & 0xffffu; };
    const uint32_t pflags = P[PW_FLAGS];
    const int out_shift = (pflags & PASS_FINAL) ? 3 : 4;
    // ---- stages ----
    std::vector<StageEval> SE(nst);
    uint32_t sign_pre = 0, sign_post = 0;
    {
      const uint32_t* S = P + PW_STAGES;
      for (uint32_t s = 0; s < nst; ++s) {
        const uint32_t hdr = S[0], sflags = (hdr >> 8) & 0xffu, rho = S[1];
        if ((hdr & 0xffu) != 3u) return fail("a stage with r != 3");
        StageEval& E = SE[s];
        E.S = S; E.k = k; E.n = n;
        for (int b = 0; b < 3; ++b) { E.rpos[b] = (rho >> (8 * b)) & 0xffu; E.sri[b] = lds_swizzle(1u << E.rpos[b]); }
        E.Qpre = (sflags & STAGE_SIGN_PRE) ? S + STAGE_HDR_WORDS : nullptr;
        E.Qpost = (sflags & STAGE_SIGN_POST) ? S + STAGE_HDR_WORDS + (E.Qpre ? SIGNQ_WORDS : 0) : nullptr;
        E.tab = S + STAGE_HDR_WORDS + (E.Qpre ? SIGNQ_WORDS : 0) + (E.Qpost ? SIGNQ_WORDS : 0);
        if (E.Qpre) sign_pre |= 1u << s;
        if (E.Qpost) sign_post |= 1u << s;
        S += hdr >> 16;
      }
    }
    const uint32_t sign_any = sign_pre | sign_post;
    const uint32_t nsign = (uint32_t)__builtin_popcount(sign_any);
    const uint32_t nrows = nst + nsign + CR_EXTRA;
    const uint32_t hb = (uint32_t)W.size();
    out.pass_off.push_back(hb);
    W.resize(hb + CH_WORDS + (size_t)nst * CS_WORDS, 0);
    W[hb + CH_NSTAGES] = nst; W[hb + CH_NROWS] = nrows; W[hb + CH_NSIGN] = nsign;
    W[hb + CH_SIGN_PRE] = sign_pre; W[hb + CH_SIGN_POST] = sign_post; W[hb + CH_NWAVES] = NW;
    for (uint32_t s = 0; s < nst; ++s) {
      const uint32_t* S = SE[s].S;
      const uint32_t sflags = (S[0] >> 8) & 0xffu;
      const uint32_t fi[4] = {S[6] & 0xffffu, S[6] >> 16, S[7] & 0xffffu, S[7] >> 16};
      uint32_t ngt = 0;
      while (ngt < 3 && fi[ngt] != 0xffffu) ++ngt;
      for (uint32_t b = ngt; b < 4; ++b) if (fi[b] != 0xffffu) return fail("gates not on register bits 0 .. ng-1");
      uint32_t* CS = W.data() + hb + CH_WORDS + (size_t)s * CS_WORDS;
      CS[CS_KIND] = ngt | ((sflags & STAGE_SIGN_PRE) ? 8u : 0u) | ((sflags & STAGE_SIGN_POST) ? 16u : 0u);
      CS[CS_CROSS] = (sflags & STAGE_CROSS_READ) ? 1u : 0u;
      for (int b = 0; b < 3; ++b) {
        CS[CS_RB + b] = S[16 + (1 << b)] << 4; CS[CS_WB + b] = S[32 + (1 << b)] << 4;
        CS[CS_MAT + b] = (fi[b] != 0xffffu ? fi[b] : 0u) * 64u;
      }
      for (int j = 0; j < 8; ++j) {
        uint32_t lr = 0, lw = 0;
        for (int b = 0; b < 3; ++b) if (j >> b & 1) { lr ^= S[16 + (1 << b)]; lw ^= S[32 + (1 << b)]; }
        if (lr != S[16 + j] || lw != S[32 + j]) return fail("slot offsets not linear in the slot number");
      }
    }
    // ---- the pass's fused matrices (byte offsets in one circuit's gate array), in stage order ----
    W[hb + CH_MAT_OFF] = (uint32_t)W.size() - hb;
    {
      uint32_t nmat = 0;
      for (uint32_t s = 0; s < nst; ++s) {
        const uint32_t* S = SE[s].S;
        const uint32_t fi[3] = {S[6] & 0xffffu, S[6] >> 16, S[7] & 0xffffu};
        for (int b = 0; b < 3; ++b) if (fi[b] != 0xffffu) { W.push_back(fi[b] * 64u); ++nmat; }
      }
      W[hb + CH_NMAT] = nmat;
    }
    // ---- ordinary tile fill / drain steps ----
    for (int m = 0; m < 3; ++m) {
      W[hb + CH_IN_STEP_N + m] = 16u << tbyte(PW_IN_PHYS, kt + m);
      W[hb + CH_FILL_STEP + m] = thalf(PW_IN_MASK, kt + m);
      W[hb + CH_DRAIN_STEP + m] = thalf(PW_OUT_MASK, kt + m);
      W[hb + CH_OUT_STEP_N + m] = P[PW_OUT_COL + kt + m] << out_shift;
    }
    // ---- direct first / last stage: offsets and eligibility exactly as build_fast_tables ----
    std::vector<uint32_t> inv_in, inv_out;
    bool ok_in = false, ok_out = false;
    uint32_t in_basis[3] = {0, 0, 0}, out_basis[3] = {0, 0, 0};
    auto in_slot_lin = [&](uint32_t u) { uint32_t o = 0; for (int j = 0; j < k; ++j) if (u >> j & 1) o ^= thalf(PW_IN_MASK, j); return o; };
    auto in_phys_u = [&](uint32_t u) { uint32_t o = 0; for (int j = 0; j < k; ++j) if (u >> j & 1) o |= 1u << tbyte(PW_IN_PHYS, j); return o; };
    auto out_slot_lin = [&](uint32_t v) { uint32_t o = 0; for (int j = 0; j < k; ++j) if (v >> j & 1) o ^= thalf(PW_OUT_MASK, j); return o; };
    auto out_phys_v = [&](uint32_t v) { uint32_t o = 0; for (int j = 0; j < k; ++j) if (v >> j & 1) o ^= P[PW_OUT_COL + j]; return o; };
    auto g_in = [&](uint32_t g, uint32_t& gslot, uint32_t& gphys) {
      gslot = gphys = 0;
      for (int m = 0; m < gbits; ++m) if (g >> m & 1) { gslot ^= thalf(PW_IN_GMASK, m); gphys |= 1u << tbyte(PW_IN_GPHYS, m); }
    };
    auto g_out = [&](uint32_t g, uint32_t& gslot, uint32_t& gphys) {
      gslot = gphys = 0;
      for (int m = 0; m < gbits; ++m) if (g >> m & 1) { gslot ^= thalf(PW_OUT_GMASK, m); gphys ^= P[PW_OUT_GCOL + m]; }
    };
    auto in_d = [&](uint32_t g, uint32_t t) {
      uint32_t gslot, gphys;
      g_in(g, gslot, gphys);
      return (in_phys_u(inv_in[(SE[0].rw(g, t) & 0xffffu) ^ gslot]) | gphys) << 4;
    };
    auto out_d = [&](uint32_t g, uint32_t t) {
      uint32_t gslot, gphys;
      g_out(g, gslot, gphys);
      return (out_phys_v(inv_out[(SE[nst - 1].rw(g, t) >> 16) ^ gslot]) ^ gphys) << out_shift;
    };
    if (nst > 0) {
      const uint32_t f0 = (SE[0].S[0] >> 8) & 0xffu, fl = (SE[nst - 1].S[0] >> 8) & 0xffu;
      if ((f0 & STAGE_FROM_HBM) && !(pflags & PASS_INIT)) {
        inv_in.assign(ksize, 0);
        for (uint32_t u = 0; u < ksize; ++u) inv_in[in_slot_lin(u)] = u;
        for (int b = 0; b < 3; ++b) in_basis[b] = in_phys_u(inv_in[W[hb + CH_WORDS + CS_RB + b] >> 4]) << 4;
        const uint32_t lanes = 1u << std::min<uint32_t>(P[PW_LO_IN], 6u);
        ok_in = P[PW_LO_IN] >= 4u && lanes <= T;
        const uint32_t runmask = lanes * 16u - 1u;
        std::vector<uint32_t> row(T);
        for (uint32_t g = 0; g < ngl && ok_in; ++g) {
          for (uint32_t t = 0; t < T; ++t) row[t] = in_d(g, t);
          for (uint32_t t0 = 0; t0 < T && ok_in; t0 += lanes) {    // every group of 2^lo_in lanes loads one aligned run
            const uint32_t base_run = row[t0] & ~runmask;
            uint64_t seen = 0;
            for (uint32_t l = 0; l < lanes; ++l) {
              if ((row[t0 + l] & ~runmask) != base_run) ok_in = false;
              seen |= 1ull << ((row[t0 + l] >> 4) & (lanes - 1u));
            }
            if (seen != (lanes == 64 ? ~0ull : ((1ull << lanes) - 1ull))) ok_in = false;
          }
        }
        for (int b = 0; b < 3; ++b) if (in_basis[b] & ((16u << std::min<uint32_t>(P[PW_LO_IN], 6u)) - 1u)) ok_in = false;
      }
      if (fl & STAGE_TO_HBM) {
        inv_out.assign(ksize, 0);
        for (uint32_t v = 0; v < ksize; ++v) inv_out[out_slot_lin(v)] = v;
        for (int b = 0; b < 3; ++b) out_basis[b] = out_phys_v(inv_out[W[hb + CH_WORDS + (size_t)(nst - 1) * CS_WORDS + CS_WB + b] >> 4]) << out_shift;
        ok_out = (n + out_shift) <= 32;
        const uint32_t lanes_o = 1u << std::min<uint32_t>(P[PW_LO_OUT], 6u);
        const uint32_t runmask = (lanes_o << out_shift) - 1u;
        if (P[PW_LO_OUT] < 4u && !(pflags & PASS_FINAL)) ok_out = false;
        if ((pflags & PASS_FINAL) && P[PW_LO_OUT] < 3u) ok_out = false;
        if (lanes_o > T) ok_out = false;
        std::vector<uint32_t> row(T);
        for (uint32_t g = 0; g < ngl && ok_out; ++g) {
          for (uint32_t t = 0; t < T; ++t) row[t] = out_d(g, t);
          for (uint32_t t0 = 0; t0 < T && ok_out; t0 += lanes_o) {
            uint32_t runs[2] = {0, 0};
            int nruns = 0;
            uint64_t seen = 0;
            for (uint32_t l = 0; l < lanes_o; ++l) {
              const uint32_t o = row[t0 + l], rb = o & ~runmask;
              int f = -1;
              for (int q = 0; q < nruns; ++q) if (runs[q] == rb) f = q;
              if (f < 0) { if (nruns == 2) { ok_out = false; break; } runs[nruns++] = rb; }
              seen |= 1ull << ((o >> out_shift) & (lanes_o - 1u));
            }
            if (nruns > ((pflags & PASS_FINAL) ? 2 : 1)) ok_out = false;
            if (seen != (lanes_o == 64 ? ~0ull : ((1ull << lanes_o) - 1ull))) ok_out = false;
          }
        }
      }
    }
    W[hb + CH_DIRECT] = (ok_in ? 1u : 0u) | (ok_out ? 2u : 0u);
    for (int b = 0; b < 3; ++b) { W[hb + CH_IN_STEP_D + b] = ok_in ? in_basis[b] : 0u; W[hb + CH_OUT_STEP_D + b] = ok_out ? out_basis[b] : 0u; }
    // ---- support of |0..0> (see build_fast_tables: FH_ZINFO) ----
    if (ok_in && i == 1 && (plan.words[plan.pass_off[0] + PW_FLAGS] & PASS_INIT) && plan.n_passes > 2) {
      const uint32_t* P0 = plan.words.data() + plan.pass_off[0];
      uint32_t zbits = 0;
      for (int m = 0; m < gbits; ++m) zbits |= 16u << ((P0[PW_OUT_GPHYS + (m >> 2)] >> (8 * (m & 3))) & 0xffu);
      uint32_t in_base = 0, in_slot = 0;
      for (uint32_t g = 0; g < ngl; ++g)
        for (uint32_t t = 0; t < T; ++t) in_base |= in_d(g, t) & zbits;
      for (int b = 0; b < 3; ++b) in_slot |= in_basis[b] & zbits;
      const uint32_t zs = in_slot & ~in_base;
      if (zs) {
        uint32_t zslots = 0;
        for (int j = 0; j < 8; ++j) {
          uint32_t off = 0;
          for (int b = 0; b < 3; ++b) if (j >> b & 1) off ^= in_basis[b];
          if (off & zs) zslots |= 1u << j;
        }
        uint32_t gmask0 = 0;
        for (int m = 0; m < gbits; ++m)
          if (zs & (16u << ((P0[PW_OUT_GPHYS + (m >> 2)] >> (8 * (m & 3))) & 0xffu))) gmask0 |= 1u << m;
        W[hb + CH_ZINFO] = zslots;
        W[out.pass_off[0] + CH_ZINFO] = gmask0;
      }
    }
    // ---- rows: point evaluators ----
    const uint32_t lo_in = P[PW_LO_IN];
    auto row_eval = [&](uint32_t row, uint32_t g, uint32_t t) -> uint32_t {
      if (row < nst) return SE[row].rw(g, t);
      if (row < nst + nsign) {
        uint32_t s = 0, seen = 0;
        for (;; ++s) if ((sign_any >> s) & 1u) { if (seen == row - nst) break; ++seen; }
        return SE[s].sg(g, t);
      }
      switch ((int)(row - nst - nsign)) {
        case CR_IN_D: return ok_in ? in_d(g, t) : 0u;
        case CR_OUT_D: return ok_out ? out_d(g, t) : 0u;
        case CR_IN_N: {
          uint32_t o = t & ((1u << lo_in) - 1u);
          for (int j = (int)lo_in; j < kt; ++j) o |= ((t >> j) & 1u) << tbyte(PW_IN_PHYS, j);
          for (int m = 0; m < gbits; ++m) o |= ((g >> m) & 1u) << tbyte(PW_IN_GPHYS, m);
          return o << 4;
        }
        case CR_OUT_N: {
          uint32_t o = 0;
          for (int j = 0; j < kt; ++j) if (t >> j & 1) o ^= P[PW_OUT_COL + j];
          for (int m = 0; m < gbits; ++m) if (g >> m & 1) o ^= P[PW_OUT_GCOL + m];
          return o << out_shift;
        }
        default: {   // CR_SLOT
          uint32_t a = 0, b2 = 0;
          for (int j = 0; j < kt; ++j) if (t >> j & 1) { a ^= thalf(PW_IN_MASK, j); b2 ^= thalf(PW_OUT_MASK, j); }
          for (int m = 0; m < gbits; ++m) if (g >> m & 1) { a ^= thalf(PW_IN_GMASK, m); b2 ^= thalf(PW_OUT_GMASK, m); }
          return a | (b2 << 16);
        }
      }
    };
    // ---- LANE / UNI / MASK ----
    const uint32_t lane_off = (uint32_t)W.size() - hb;
    W.resize(W.size() + (size_t)nrows * 64, 0);
    const uint32_t uni_off = (uint32_t)W.size() - hb;
    W.resize(W.size() + (size_t)ngl * nrows * NW, 0);
    const uint32_t mask_off = (uint32_t)W.size() - hb;
    W.resize(W.size() + (size_t)ngl * (nsign ? nsign : 1u) * NW, 0);
    W[hb + CH_LANE_OFF] = lane_off; W[hb + CH_UNI_OFF] = uni_off; W[hb + CH_MASK_OFF] = mask_off;
    for (uint32_t row = 0; row < nrows; ++row) {
      const bool is_sign = row >= nst && row < nst + nsign;
      const uint32_t f00 = row_eval(row, 0, 0);
      uint32_t* LANE = W.data() + hb + lane_off + (size_t)row * 64;
      for (uint32_t l = 0; l < 64; ++l) LANE[l] = l < T ? row_eval(row, 0, l) : 0u;
      for (uint32_t g = 0; g < ngl; ++g)
        for (uint32_t w = 0; w < NW; ++w) {
          const uint32_t fgw = row_eval(row, g, 64u * w);
          W[hb + uni_off + ((size_t)g * nrows + row) * NW + w] = fgw ^ f00;
          if (is_sign) {
            uint32_t mpre = 0, mpost = 0;
            for (int b = 0; b < 6 && b < kt; ++b) {
              const uint32_t d = row_eval(row, g, 64u * w + (1u << b)) ^ LANE[1u << b] ^ fgw ^ f00;
              mpre |= (d & 1u) << b;
              mpost |= ((d >> 16) & 1u) << b;
            }
            W[hb + mask_off + ((size_t)g * nsign + (row - nst)) * NW + w] = mpre | (mpost << 8);
          }
        }
      // check: every (g, t) when there are at most 2^22 of them, else all t of 64 tile rows spread over the range
      const uint32_t gstep = ((size_t)ngl * T <= ((size_t)1 << 22)) ? 1u : ngl / 64u;
      for (uint32_t g = 0; g < ngl; g += gstep)
        for (uint32_t t = 0; t < T; ++t) {
          const uint32_t l = t & 63u, w = t >> 6;
          uint32_t v = LANE[l] ^ W[hb + uni_off + ((size_t)g * nrows + row) * NW + w];
          if (is_sign) {
            const uint32_t mk = W[hb + mask_off + ((size_t)g * nsign + (row - nst)) * NW + w];
            if (par32(l & (mk & 0xffu))) v ^= 0x0000ffffu;
            if (par32(l & (mk >> 8))) v ^= 0xffff0000u;
          }
          if (v != row_eval(row, g, t)) return fail("a table word is not affine in (tile row, thread): pass " + std::to_string(i) + " row " + std::to_string(row));
        }
    }
    out.max_rows = std::max(out.max_rows, (int)nrows);
    out.max_sign = std::max(out.max_sign, (int)nsign);
    out.max_stages = std::max(out.max_stages, (int)nst);
  }
  if (W.size() >= (size_t)0x7fffffffu) return fail("tables too large");
  return true;
}

}  // namespace bornvi
