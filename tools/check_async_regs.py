#!/usr/bin/env python3
"""Build-time check of the hand-scheduled vector memory ops in circuit_pass_fast_kernel.

The kernel issues its next-tile prefetch with inline-asm `global_load_dwordx4` (the compiler does not track them) and
waits with hand-written `s_waitcnt vmcnt(N)`.  The hardware does not interlock VGPR reads against pending loads, so
between a prefetch load and the next hand-written wait NO instruction may read or write the destination registers --
in particular no register copy the compiler inserts to merge live ranges (seen once: 34 v_mov right behind the
prefetch; the second tile of every workgroup was computed from stale registers).  This script compiles
kernels_circuit.hip to assembly and checks exactly that, in linear code order, for every instantiation.
Exit code 0 = clean."""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(REPO, "tensornetworks_amd", "csrc")
# (source file, mangled-name prefix of the kernels whose prefetch is hand-scheduled)
TARGETS = [(os.path.join(CSRC, "kernels_circuit.hip"), "_ZN6bornvi24circuit_pass_fast_kernel"),
           (os.path.join(CSRC, "kernels_circuit8.hip"), "_ZN6bornvi22circuit_pass_r3_kernel")]
SRC = TARGETS[0][0]


def regs_of(tok):
    """v12 -> {12}; v[8:11] -> {8..11}"""
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", tok):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", tok):
        out.add(int(m.group(1)))
    return out


def check(asm_text, prefix="_ZN6bornvi24circuit_pass_fast_kernel"):
    """Linear scan of each instantiation: a queue of outstanding vector-memory ops in issue order (hand-written
    ones inside ASMSTART/ASMEND and the compiler's own), drained by every s_waitcnt vmcnt(N) down to its N youngest.
    The scan follows the file order, i.e. the loop body once from the load site to the loop latch and on through the
    top of the next trip as the compiler laid it out after it; branches are not followed.  (A path-exact version
    needs the loop invariants the vmcnt counts rest on -- "a pass with a direct last stage has more than one stage",
    "every stage kind is one of the twenty" -- and flags infeasible paths without them; the multi-trip parity test
    tests/test_gpu_circuit.py::test_persistent_tile_loop is the check on the hardware.)"""
    problems = []
    kernels = re.split(r"\n(?=" + re.escape(prefix) + r"[^\n]*:\s*;)", asm_text)
    n_sites = 0
    for chunk in kernels[1:]:
        name = chunk.split(":", 1)[0][-60:]
        lines = chunk.split("s_endpgm")[0].split("\n")
        in_asm = False
        queue = []            # [(line, set(dst regs))] oldest first
        for ln, line in enumerate(lines):
            s = line.strip()
            if s.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if s.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not s or s.startswith(";") or s.startswith("."):
                continue
            code = s.split(";")[0].strip()
            m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", code)
            if m:
                keep = int(m.group(1))
                queue = queue[len(queue) - keep:] if keep else []
                continue
            if code.startswith("s_waitcnt") and "vmcnt" not in code:
                continue
            inflight = set().union(*[d for _, d in queue]) if queue else set()
            if not in_asm:
                touched = regs_of(code) & inflight
                if touched:
                    src = [l for l, d in queue if d & touched][0]
                    problems.append(f"{name}: line {ln}: `{code}` touches in-flight v{sorted(touched)[:4]} (load at line {src})")
            if re.match(r"(global|scratch|buffer|flat)_(load|store|atomic)", code):
                dst = regs_of(code.split(",")[0]) if "_load" in code else set()
                if in_asm and code.startswith("global_load_dwordx4"):
                    n_sites += 1
                elif not in_asm:
                    dst = set()       # the compiler waits for its own loads itself
                queue.append((ln, dst))
    return n_sites, problems


def main():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    rc = 0
    only = sys.argv[1] if len(sys.argv) > 1 else None
    with tempfile.TemporaryDirectory() as td:
        for src, prefix in TARGETS:
            if not os.path.exists(src) or (only and only not in src):
                continue
            out = os.path.join(td, "kc.s")
            subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(REPO, "include"),
                            "-I" + os.path.dirname(src), "-S", "--cuda-device-only", src, "-o", out], check=True,
                           stderr=subprocess.DEVNULL)
            n_sites, problems = check(open(out).read(), prefix)
            print(f"[check_async_regs] {os.path.basename(src)}: {n_sites} prefetch loads checked, {len(problems)} problem(s)")
            for p in problems[:20]:
                print("   ", p)
            if problems or n_sites == 0:
                rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
