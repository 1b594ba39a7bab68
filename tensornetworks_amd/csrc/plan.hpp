// Circuit execution planner (host side).
//
// Turns an ansatz (gate order of quantum_born_machine.py:57-128) into a small program
// for the LDS-tiled statevector kernel:
//
//   gates --fuse--> ops (U1 = fused one-qubit unitary, CX, CZ)
//         --passes--> each pass makes `k` of the n wires tile-local (a 2^k-amplitude
//                     tile lives in one workgroup's LDS); ops whose non-diagonal target is
//                     local run in that pass; the tile is re-laid-out on the way back to
//                     HBM so that the next pass finds ITS wires local and its lowest
//                     physical bits contiguous (coalesced 1 KiB runs);
//         --stages--> inside a pass every thread keeps 2^r amplitudes in registers
//                     (r register wires); one LDS round trip per stage.
//
// The serialised plan is a flat uint32 array read by the kernels with wave-uniform
// (scalar) loads.  Layout constants below are shared with kernels_circuit.hip and with the
// Python plan emulator in tests/.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace bornvi {

enum GateKind : int { G_H = 0, G_RX = 1, G_RY = 2, G_RZ = 3, G_CNOT = 4, G_CZ = 5 };

struct Gate {
  int kind;
  int w0;     // wire (1q) or control / first wire
  int w1;     // target / second wire, -1 for 1q gates
  int param;  // parameter index or -1
};

int num_params(int ansatz, int n, int layers);
// false if ansatz is unknown
bool build_gate_list(int ansatz, int n, int layers, std::vector<Gate>& out);

// ---- serialised plan layout (uint32 words) --------------------------------------------------
constexpr uint32_t PLAN_MAGIC = 0x424e5649u;  // "BNVI"
enum PlanHeader : int {
  PH_MAGIC = 0, PH_N, PH_K, PH_NPASSES, PH_NFUSED, PH_NPARAMS, PH_OFF_FUSED, PH_OFF_PASSTAB,
  PH_TOTAL, PH_THREADS, PH_R, PH_NGATES, PH_SIZE = 16
};
// fused one-qubit gate: FUSED_WORDS words each
//   [0] wire  [1] n_elem  [2+2e] kind_e  [3+2e] param_e   (e < FUSED_MAX_ELEMS), applied e = 0 first
constexpr int FUSED_MAX_ELEMS = 4;
constexpr int FUSED_WORDS = 2 + 2 * FUSED_MAX_ELEMS;
// pass descriptor: a 64-word header the kernel fetches with four wide scalar loads.  Bit positions are
// packed one per byte (16 bytes = 4 words per table, so k <= 16 and n - k <= 16), LDS slot masks one per
// 16 bits (8 words per table).
//
// Tile <-> HBM mapping.  Load: element u of the load enumeration (bit j of u = LDS/tile bit j of the pass's
// input frame) is read from phys-in address  sum_j bit_j(u) << IN_PHYS[j]  |  sum_m bit_m(g) << IN_GPHYS[m]
// and stored at the SWIZZLED LDS slot  xor_j bit_j(u) IN_MASK[j]  ^  xor_m bit_m(g) IN_GMASK[m].
// Without folded CNOTs IN_MASK[j] = lds_swizzle(1 << j); CNOTs at the head of the pass (targets tile-local)
// are GF(2)-linear maps of the tile index and are folded into these masks -- they cost nothing.
// Store: element v of the out enumeration is read from slot  xor_j bit_j(v) OUT_MASK[j] ^ xor_m bit_m(g) OUT_GMASK[m]
// (CNOTs at the tail of the pass folded in) and written to  sum_j bit_j(v) << OUT_PHYS[j] | sum_m bit_m(g) << OUT_GPHYS[m]
// -- in general to the GF(2)-linear address given by the PW_OUT_COL / PW_OUT_GCOL columns (see there).
enum PassWords : int {
  PW_FLAGS = 0, PW_K, PW_N, PW_NSTAGES, PW_LO_IN, PW_LO_OUT, PW_THREADS, PW_RESERVED,
  PW_IN_PHYS = 8,       // [16 bytes]
  PW_IN_GPHYS = 12,     // [16 bytes]
  PW_OUT_PHYS = 16,     // [16 bytes]
  PW_OUT_GPHYS = 20,    // [16 bytes]
  PW_IN_MASK = 24,      // [16 x 16 bit]
  PW_IN_GMASK = 32,     // [16 x 16 bit]
  PW_OUT_MASK = 40,     // [16 x 16 bit]
  PW_OUT_GMASK = 48,    // [16 x 16 bit]
  PW_HEADER_WORDS = 64,
  PW_WIRE_OF_LDS = 64,  // [32 words] wire held by LDS bit j        (informational / emulator)
  PW_WIRE_OF_G = 96,    // [32 words] wire held by workgroup bit m  (informational / emulator)
  PW_MATS = 128,        // [MAX_STAGES * 2 words] fused gate index of (stage s, register bit i), 16 bits each:
                        //   word PW_MATS + 2 s + (i >> 1); the workgroup copies these matrices of ITS circuit
                        //   into LDS (after the tile) while the tile loads are in flight
  PW_OUT_COL = 192,     // [16 words] phys-out address = xor_j bit_j(v) OUT_COL[j] ^ xor_m bit_m(g) OUT_GCOL[m]:
  PW_OUT_GCOL = 208,    // [16 words]   one-hot columns (1 << OUT_PHYS[j]) except in the last pass of a circuit, where
                        //   the CNOTs that END the circuit (only diagonal gates or CNOTs after them) are not
                        //   executed at all: |psi|^2 is written to the permuted outcome index instead
  PW_STAGES = 224
};
constexpr int MAX_STAGES = 32;          // stages per pass
constexpr int STAGE_MATS_BYTES = 4 * 64; // LDS bytes of one stage's four 2x2 complex matrices
constexpr uint32_t PASS_INIT = 1u;   // tile starts as |0...0> (no HBM read)
constexpr uint32_t PASS_FINAL = 2u;  // epilogue writes |psi|^2 in canonical order
constexpr uint32_t PASS_FINAL_STATE = 4u;  // epilogue writes the state itself in canonical order
// stage (fixed form, see plan.cpp): STAGE_HDR_WORDS words, optional sign payloads, then the base table
//   [0]      r | flags << 8 | nwords << 16
//   [1]      rho: 4 bytes, LDS bit position of register bit t
//   [2..5]   tpos: 16 bytes, LDS bit position receiving thread-id bit j   (emulator / generic kernel)
//   [6..7]   fused gate index of register bit t (16 bits each, 0xffff = no gate)
//   [8..11]  Bpre[t]:  parity(e & Bpre[t]) flips register bit t of the LDS READ index
//   [12..15] Bpost[t]: same for the LDS WRITE index        (e = extended index: LDS bits, then workgroup bits)
//   [16..31] load_off[16]:  swizzled LDS offset register slot j is read from
//   [32..47] store_off[16]: swizzled LDS offset register slot j is written to
//   [48..63] BpreF[p]: parity(e & BpreF[p]) flips LDS bit p of the READ index (p not a register position): phase-0 CNOTs
//            whose target is a thread-held wire -- a GF(2)-linear read map costs nothing whichever bits it moves
//   then SIGNQ_WORDS words if STAGE_SIGN_PRE, then SIGNQ_WORDS words if STAGE_SIGN_POST,
//   then 2^(k-r) words: thread t's  lds_swizzle(base_t) | base_t << 16  (base_t = t's bits deposited at the
//   non-register LDS positions) -- index arithmetic done once by the planner instead of per stage per thread
constexpr int STAGE_HDR_WORDS = 64;
constexpr uint32_t STAGE_SIGN_PRE = 1u;
constexpr uint32_t STAGE_SIGN_POST = 2u;
// thread-bit order of the stage allows the fast kernel to take the stage's amplitudes straight from HBM (first
// stage of a pass) / to send them straight to HBM (last stage) with coalesced accesses; the generic kernel
// ignores these flags
constexpr uint32_t STAGE_FROM_HBM = 4u;
constexpr uint32_t STAGE_TO_HBM = 8u;
// the stage's READ map takes amplitudes from other threads' groups (BpreF / slot offsets outside the register positions):
// every thread must have finished reading before any thread writes -- the kernels put a workgroup barrier between the
// stage's reads and its write-back
constexpr uint32_t STAGE_CROSS_READ = 16u;
// sign payload (product of CZ gates = (-1)^{q(x)}, q a quadratic form over the extended index bits):
//   [0..31] U rows (upper-triangular adjacency), [32..47] m_j (bilinear masks of the 16 slot
//   offsets), [48] q bits of the slot offsets
constexpr int SIGNQ_WORDS = 49;

// LDS index swizzle shared by planner and kernel: xor-fold the upper nibbles into the low nibble.
// Linear over GF(2), its own inverse, keeps bits >= 4.
#if defined(__HIPCC__)
#define BORNVI_HD __host__ __device__
#else
#define BORNVI_HD
#endif
BORNVI_HD inline uint32_t lds_swizzle(uint32_t l) { return l ^ (((l >> 4) ^ (l >> 8) ^ (l >> 12)) & 15u); }

constexpr size_t FAST_TABLE_MAX_BYTES = (size_t)256 << 20;
constexpr size_t MAX_LDS_BYTES = 160 * 1024;   // per CU and per workgroup on gfx950

struct PlanOptions {
  int kmax = 13;     // largest tile (2^13 complex128 = 128 KiB of LDS): a state of n <= kmax qubits is ONE tile
  int kmulti = 0;    // tile bits when the state needs several tiles (n > kmax).  0 = by measurement on MI355X
                     // (DESIGN.md 4.1): 2^13 (one 512-thread workgroup per CU, the large-tile instantiation of the
                     // fast kernel) wherever the fast kernel can run it, else 2^11 (make_plan)
  int r = 3;         // register wires per stage: 3 = 2^3 amplitudes per thread (circuit_pass_r3_kernel: <= 128 VGPRs, four waves per
                     // SIMD; the default: faster than or level with the other for every n measured), 4 = 2^4 per
                     // thread (circuit_pass_fast_kernel, <= 256 VGPRs, two waves per SIMD; also the fallback where a plan is not
                     // eligible for the first)
  int lo = 6;        // contiguous low physical bits per HBM access (2^6 * 16 B = 1 KiB)
  int max_threads = 1024;
  int contig_out = 1;      // physical layout of the buffer a pass writes: its OWN local wires on the low k bits (the wires it shares
                           // with the next pass lowest), so that a tile is stored as one contiguous 2^k-amplitude block and the
                           // next pass reads runs of 2^(shared wires).  0: only the `lo` shared wires are low, the rest in wire
                           // order (both sides move 1-KiB runs).  Measured on the MI355X (tools/probes/tile_copy_probe.hip, the
                           // tile traffic of a pass alone): both sides in 1-KiB runs 4.7 TB/s, contiguous stores + 1-KiB-run
                           // loads 5.4 TB/s, the reverse 5.2 TB/s, both contiguous 5.8 TB/s.
  int read_map = -1;       // phase-0 CNOTs may target thread-held wires (general GF(2) read map, STAGE_CROSS_READ): fewer stages,
                           // but such a stage needs a barrier between its reads and its write-back.  Measured on the MI355X
                           // (n = 16 / 20): with 16 amplitudes per thread (8 waves in step per CU) the barrier costs more than
                           // the stages saved (+3.7 % / +2.8 %); with 8 per thread (16 waves) it pays (-2.6 % / -5.7 %: 44 -> 37
                           // and 70 -> 58 stages).  -1 = by that measurement (on iff r == 3), 0 / 1 = forced (option "read_map")
  bool use_read_map() const { return read_map < 0 ? r == 3 : read_map != 0; }
};

struct Plan {
  int n = 0, k = 0, r = 0, n_passes = 0, n_fused = 0, n_params = 0, threads = 0, n_gates = 0;
  int max_stages = 0;   // most stages in any one pass (sizes the LDS matrix area)
  std::vector<uint32_t> words;
  std::vector<uint32_t> pass_off;  // word offset of each pass descriptor
  std::vector<int> param_first_pass;   // first pass whose matrices depend on parameter p: a circuit that differs from
                                       // the base circuit only in p has the base circuit's state before that pass
  // tile + the matrices of the longest pass, rounded up to 512 bytes
  size_t lds_bytes() const { return (size_t(1) << k) * 16 + (((size_t)(max_stages > 0 ? max_stages : 1) * STAGE_MATS_BYTES + 511) / 512) * 512; }
  // fast kernel: the above, then one tile row of the RW (and SG) stage tables of the longest pass
  // LDS of the fast kernel: tile | matrices of the current tile | table rows | matrices of the next tile.
  // `tab_rows` (FastTables::max_tab_rows) = most RW rows + SG rows (sign stages only) any pass needs; four more
  // rows hold per-thread words (matrix piece, first / last stage HBM offsets).
  size_t fast_mats_bytes() const { return (size_t)(max_stages > 0 ? max_stages : 1) * STAGE_MATS_BYTES; }
  size_t fast_lds_tab_off() const { return (size_t(1) << k) * 16 + fast_mats_bytes(); }
  size_t fast_tab_bytes(int tab_rows) const { return ((size_t)(tab_rows + 4) * ((size_t)1 << (k - 4)) * 4 + 15) / 16 * 16; }
  size_t fast_lds_mats2_off(int tab_rows) const { return fast_lds_tab_off() + fast_tab_bytes(tab_rows); }
  size_t fast_lds_bytes(int tab_rows) const { return fast_lds_mats2_off(tab_rows) + fast_mats_bytes(); }
};

// ---- fast-path tables (kernels_circuit.hip: circuit_pass_fast_kernel) ---------------------------
// Derived from the serialised plan above (same semantics, evaluated once on the host): for every stage of
// every pass and every (tile g, thread t) the swizzled LDS slot the thread's 16-amplitude group is read from /
// written to with ALL index-dependent CNOT flips folded in, and -- for stages carrying CZ products -- the
// 16 sign bits of its slots.  The kernel then does no index arithmetic per stage beyond one table load.
// Layout of `words` (uint32):
//   pass header, FH_WORDS words:  [FH_NSTAGES] [FH_RW_BASE] [FH_SG_BASE] [FH_SIGN_PRE] [FH_SIGN_POST]
//       RW_BASE / SG_BASE: word offsets (from the start of `words`) of the tables of stage 0; stage s is
//       2^(n-4) words further per stage; SIGN_PRE / SIGN_POST: bit s set = stage s applies that sign
//   stage header s at  pass header + FH_WORDS + s * FS_WORDS:
//       [FS_FI01] [FS_FI23] fused gate of register bit i (16 bits each, 0xffff = none)
//       [FS_KIND] number of fused gates (they sit on register bits 0 .. ng-1) | pre sign << 3 | post sign << 4
//       [FS_RB .. +3] byte offset xor-ed into the LDS READ address for register bit i of the slot number
//       [FS_WB .. +3] same for the WRITE address   (slot offsets are GF(2)-linear in the slot number)
//   tables: RW[s][g][t] = read slot | write slot << 16;  SG[s][g][t] = pre signs | post signs << 16
//   direct HBM <-> register stages: [FH_IN_TAB] / [FH_OUT_TAB] word offset of a [g][t] table (0 = none) with the
//       BYTE offset, inside one state resp. one probability vector, of slot 0 of the thread's group in the buffer the
//       pass reads (first stage) resp. writes (last stage); slot j is at  offset ^ xor of the FH_IN_BASIS /
//       FH_OUT_BASIS words of the bits of j  (both address maps are GF(2)-linear)
enum FastHeader : int { FH_NSTAGES = 0, FH_RW_BASE, FH_SG_BASE, FH_SIGN_PRE, FH_SIGN_POST, FH_IN_TAB, FH_OUT_TAB,
                        FH_ZINFO = 7,   // support of |0..0>: see build_fast_tables (INIT pass: tiles nobody reads; next pass: zero slots)
                        FH_IN_BASIS = 8, FH_OUT_BASIS = 12, FH_WORDS = 16 };
enum FastStage : int { FS_FI01 = 0, FS_FI23, FS_RB = 2, FS_WB = 6, FS_KIND = 10, FS_CROSS = 11 /* STAGE_CROSS_READ */, FS_WORDS = 16 };

// The stage kinds circuit_pass_fast_kernel has a specialised body for (stage_dispatch's switch): 0..4 fused gates on
// register bits 0..ng-1, with or without either CZ sign product.  A stage of any other kind would issue NO tile stores
// and break the kernel's hand-counted vmcnt waits, so build_fast_tables refuses the whole plan instead (the generic
// kernel then runs it).
constexpr bool fast_stage_kind_supported(uint32_t kind) { return (kind & 7u) <= 4u && (kind >> 5) == 0u; }

struct FastTables {
  int r = 4;                        // register wires of the plan the tables were built for (slots per thread = 2^r)
  std::vector<uint32_t> words;
  std::vector<uint32_t> pass_off;   // word offset of each pass's header
  bool any_sign = false;            // some stage carries a CZ sign product (the SG tables exist)
  int max_tab_rows = 1;             // most (stages + sign stages) of any pass: table rows a workgroup keeps in LDS
};
// false (tables empty) when the plan is not eligible: tiles smaller than 2^10 (2^9 with r = 3), a pass with more 16-byte
// matrix pieces (4 r per stage) than threads, or tables above `max_bytes`; the generic kernel then runs the plan.
// With r = 3 every table has 2^(n-3) words per stage, a thread has 8 slots and 3 basis offsets (same layout otherwise).
bool build_fast_tables(const Plan& plan, size_t max_bytes, FastTables& out);

// ---- compact tables (kernels_circuit8.hip: circuit_pass_r3_kernel) ---------------------------------
// Every per-(tile row g, thread t) word of the fast tables is GF(2)-affine in (g, t): LDS slots, HBM byte offsets and
// the slot-relative CZ sign bits are xor-sums of per-bit columns.  So  word(g, t) = LANE[row][t & 63] ^ UNI[g][row][t >> 6]:
// 64 lane words per row (the same for every wave and tile row) and one wave-uniform word per (tile row, row, wave) --
// a few KiB of LDS per workgroup instead of one word per thread and row (which at 8 amplitudes per thread and
// 1024 threads would not fit beside a 2^13 tile).  The CZ sign common to a thread's slots is a QUADRATIC form of
// (g, t): q(l + h) = q(l) + q(h) + B(l, h), l = lane part, h = wave and tile-row part; B(l, h) = parity(lane & M[g][w])
// with a 6-bit mask M per (tile row, sign row, wave): word ^= 0xffff (pre) / 0xffff0000 (post) where that parity is odd.
// Layout of `words` per pass, at pass_off[i]:
//   header CH_WORDS words, then nstages stage headers of CS_WORDS words, then
//   MAT[CH_NMAT]: byte offsets, in one circuit's gate array, of the fused matrices the pass uses (the kernel reads them with
//   scalar loads, 64 bytes each, and touches the next tile's a trip ahead)
//   LANE[nrows][64], UNI[2^(n-k)][nrows][nwaves], MASK[2^(n-k)][nsign][nwaves] (pre mask | post mask << 8)
// rows: 0 .. nstages-1: read slot | write slot << 16 of the stage;  then one row per stage carrying a sign (in stage
// order): pre sign bits | post sign bits << 16;  then CR_IN_D, CR_IN_N, CR_OUT_D, CR_OUT_N, CR_SLOT.
enum CompactHeader : int {
  CH_NSTAGES = 0, CH_NROWS, CH_NSIGN, CH_SIGN_PRE, CH_SIGN_POST,
  CH_DIRECT,        // bit 0: the first stage can take its amplitudes straight from HBM (row CR_IN_D), bit 1: the last stage can store straight to HBM
  CH_ZINFO,         // support of |0..0>, as FH_ZINFO
  CH_NWAVES, CH_MAT_OFF, CH_LANE_OFF, CH_UNI_OFF, CH_MASK_OFF,     // offsets from the pass's header
  CH_NMAT = 30,         // fused matrices of the pass (entries of MAT)
  CH_IN_STEP_D = 12,    // [3] byte offsets xor-ed into a thread's CR_IN_D word for bit m of the element number (= slot number of the first stage)
  CH_IN_STEP_N = 15,    // [3] the same for the ordinary tile fill (row CR_IN_N): 16 << phys-in position of enumeration bit k-3+m
  CH_FILL_STEP = 18,    // [3] LDS slot masks of those enumeration bits (tile fill; thread part: low half of row CR_SLOT)
  CH_DRAIN_STEP = 21,   // [3] LDS slot masks of the top 3 out-enumeration bits (tile drain; thread part: high half of row CR_SLOT)
  CH_OUT_STEP_D = 24,   // [3] byte offsets of the slot-number bits of the last stage (row CR_OUT_D)
  CH_OUT_STEP_N = 27,   // [3] byte offsets of the top 3 out-enumeration bits (row CR_OUT_N)
  CH_WORDS = 32
};
// stage header: kind (fused gates on register bits 0 .. ng-1 | pre sign << 3 | post sign << 4), cross-read flag, byte offsets
// xor-ed into the LDS read / write address for the bits of the slot number, byte offset of register bit i's matrix in one
// circuit's gate array
enum CompactStage : int { CS_KIND = 0, CS_CROSS, CS_RB = 2, CS_WB = 5, CS_MAT = 8, CS_WORDS = 12 };
enum CompactRow : int { CR_IN_D = 0, CR_IN_N, CR_OUT_D, CR_OUT_N, CR_SLOT, CR_EXTRA = 5 };   // + nstages + nsign
struct CompactTables {
  std::vector<uint32_t> words;
  std::vector<uint32_t> pass_off;
  int max_rows = 0, max_sign = 0, max_stages = 0;
  // LDS of circuit_pass_r3_kernel: tile | LANE | UNI (one tile row) | MASK | wave sums
  size_t lds_bytes(int k) const {
    const size_t nw = std::max<size_t>(((size_t)1 << (k - 3)) / 64, 1);
    return ((size_t)16 << k) + (size_t)max_rows * 64 * 4 + (size_t)max_rows * nw * 4 + (size_t)(max_sign > 0 ? max_sign : 1) * nw * 4 + 8 + nw * 8 + 64;   // (+ the waves' partial sums of the fused dot)
  }
};
// false (with msg) when the plan is not eligible (r != 3, tiles below 2^9, more matrix pieces than threads) or when a table
// word turns out not to be affine in (g, t) -- every word is checked against its point evaluation (all (g, t) up to
// 2^22 of them per row, a fixed sample beyond)
bool build_compact_tables(const Plan& plan, CompactTables& out, std::string& msg);

// Returns false (with msg) on unsupported sizes.
bool make_plan(int ansatz, int n, int layers, const PlanOptions& opt, Plan& out, std::string& msg);
// Program applying one shared 2x2 matrix to every bit of a canonical-order vector (state in,
// state out); used by the matrix-free Stein mat-vec (K_base = M^{(x) n}).
bool make_kron_plan(int n, const PlanOptions& opt, Plan& out, std::string& msg);

}  // namespace bornvi
