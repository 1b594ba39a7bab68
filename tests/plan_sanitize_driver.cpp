// Host-only driver for the AddressSanitizer / UBSan build of the circuit planner (tests/test_guard_rails.py):
// plans a sweep of (ansatz, n, layers, tile bits), builds the fast-path tables, and prints one line per case with
// FNV-1a hashes of the plan words and of the table words.  The test compares the lines with what the production
// library returns for the same cases; the sanitizers abort on any out-of-bounds access or undefined behaviour in
// plan.cpp (985 lines of index arithmetic that feed addresses to the GPU kernels).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "plan.hpp"

static uint64_t fnv(const std::vector<uint32_t>& w) {
  uint64_t h = 1469598103934665603ull;
  for (uint32_t v : w)
    for (int b = 0; b < 4; ++b) { h ^= (v >> (8 * b)) & 0xffu; h *= 1099511628211ull; }
  return h;
}

int main(int argc, char** argv) {
  // argv: quadruples "ansatz n layers tile_bits" (ansatz -1 = Kronecker mat-vec plan)
  for (int i = 1; i + 3 < argc; i += 4) {
    const int ansatz = atoi(argv[i]), n = atoi(argv[i + 1]), L = atoi(argv[i + 2]), kb = atoi(argv[i + 3]);
    // (kb as the library's describe functions take it: bits 0-7 tile bits, bit 8 read map, bit 9 the 3-register-wire plan)
    bornvi::PlanOptions opt;
    opt.read_map = (kb & 0x100) ? 1 : 0;
    opt.r = (kb & 0x200) ? 3 : 4;
    if (opt.r == 4) opt.max_threads = 512;
    if ((kb & 0xff) > 0) { opt.kmax = kb & 0xff; opt.kmulti = kb & 0xff; }
    bornvi::Plan p;
    std::string msg;
    const bool ok = ansatz == -1 ? bornvi::make_kron_plan(n, opt, p, msg) : bornvi::make_plan(ansatz, n, L, opt, p, msg);
    if (!ok) { printf("%d %d %d %d unsupported\n", ansatz, n, L, kb); continue; }
    bornvi::FastTables ft;
    const bool fast = bornvi::build_fast_tables(p, bornvi::FAST_TABLE_MAX_BYTES, ft);
    bool kinds_ok = true;
    if (fast)
      for (size_t q = 0; q < ft.pass_off.size(); ++q) {
        const uint32_t* F = ft.words.data() + ft.pass_off[q];
        for (uint32_t s = 0; s < F[bornvi::FH_NSTAGES]; ++s)
          kinds_ok = kinds_ok && bornvi::fast_stage_kind_supported(F[bornvi::FH_WORDS + s * bornvi::FS_WORDS + bornvi::FS_KIND]);
      }
    // the compact tables of the 8-amplitude kernel (every word checked against its point evaluation by the builder)
    bornvi::CompactTables ct;
    std::string cmsg;
    const bool compact = p.r == 3 && bornvi::build_compact_tables(p, ct, cmsg) && ct.lds_bytes(p.k) <= bornvi::MAX_LDS_BYTES;
    printf("%d %d %d %d plan %zu %016llx fast %zu %016llx kinds %d compact %zu %016llx\n", ansatz, n, L, kb, p.words.size(),
           (unsigned long long)fnv(p.words), fast ? ft.words.size() : (size_t)0,
           (unsigned long long)(fast ? fnv(ft.words) : 0ull), kinds_ok ? 1 : 0, compact ? ct.words.size() : (size_t)0,
           (unsigned long long)(compact ? fnv(ct.words) : 0ull));
  }
  return 0;
}
