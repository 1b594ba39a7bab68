#!/usr/bin/env python3
"""Builds libbornvi_hip.so for gfx950 (MI355X) with hipcc; in-tree output next to the package.
One object per source file (compiled in parallel, rebuilt only when the file or a header changed), then one link."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
REPO = os.path.dirname(PKG)
SOURCES = ["api.hip", "kernels_circuit.hip", "kernels_circuit8.hip", "kernels_stein.hip", "kernels_batched.hip", "kernels_adjoint.hip", "plan.cpp"]
HEADERS = [os.path.join(HERE, h) for h in ("plan.hpp", "kernels.hpp", "circuit_dev.hpp", "exports.map")] + [os.path.join(REPO, "include", "bornvi.h")]
OUT = os.path.join(PKG, "libbornvi_hip.so")
OBJ = os.path.join(HERE, "_obj")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fvisibility=hidden", "-I" + os.path.join(REPO, "include"), "-I" + HERE,
         "-Rpass-analysis=kernel-resource-usage"]

# Register-allocation guard rails.  The hot kernels sit a few registers below a cliff: a small source edit has more than
# once made the compiler spill hundreds of VGPRs to scratch (the symmetric contraction went from 2.6 to 6.6 ms that way,
# with identical results), and in the circuit engine a scratch access inside the tile loop would also break the
# hand-counted vmcnt waits.  The build fails when a kernel leaves its budget: (max scratch bytes per lane, max VGPR spills).
RESOURCE_BUDGET = {
    "quadform_sym_kernel": (0, 0),
    "quadform_batched_kernel": (0, 0),
    "quadform_kernel": (0, 0),
    "gram_mfma_kernel": (0, 0),
    "circuit_pass_fast_kernelILb0": (20, 0),      # (20 bytes: spilled SGPRs of the set-up code, outside the tile loop)
    "circuit_pass_r3_kernelILb0": (0, 0),         # 8 amplitudes per thread: <= 128 VGPRs (four waves per SIMD) without scratch
    "circuit_pass_r3_kernelILb1": (24, 4),        # its fused-dot instantiation (last pass only): 8 weights more per thread; the few
                                                  # spilled address words are reloaded at the end of a trip, not inside the stages
}


def _stale(target, deps):
    return not os.path.exists(target) or any(os.path.getmtime(target) < os.path.getmtime(d) for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = [os.path.join(HERE, s) for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    os.makedirs(OBJ, exist_ok=True)
    objs, jobs = [], []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s) + ".o")
        objs.append(o)
        if force or _stale(o, [s] + HEADERS + [os.path.abspath(__file__)]):
            jobs.append([hipcc] + FLAGS + ["-c", s, "-o", o])
    if not force and not _stale(OUT, srcs + HEADERS):     # (the objects do not travel to the GPU box; the .so does)
        if verbose:
            print(f"[bornvi] {OUT} is up to date")
        return OUT
    if not os.path.exists(hipcc):
        raise RuntimeError(f"{hipcc} not found and {OUT} is out of date")

    def run(cmd):
        if verbose:
            print("[bornvi]", " ".join(c for c in cmd if not c.startswith("-Rpass")), flush=True)
        if "-c" in cmd:          # keep the compiler's resource remarks next to the object
            r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
            with open(cmd[cmd.index("-o") + 1] + ".remarks", "w") as f:
                f.write(r.stderr)
            if r.returncode != 0:
                sys.stderr.write("\n".join(l for l in r.stderr.splitlines() if "remark:" not in l and not l.startswith(" ")) + "\n")
                raise subprocess.CalledProcessError(r.returncode, cmd)
        else:
            subprocess.run(cmd, check=True)
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    check_resources(objs, verbose)
    run([hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-Wl,--version-script=" + os.path.join(HERE, "exports.map"), "-o", OUT] + objs)
    return OUT


def kernel_resources(objs):
    """{mangled kernel name: {"scratch": bytes per lane, "vgpr_spill": n, "vgprs": n, "agprs": n, "sgpr_spill": n}} from
    the remarks files written beside the objects."""
    import re
    out = {}
    for o in objs:
        path = o + ".remarks"
        if not os.path.exists(path):
            continue
        cur = None
        for line in open(path):
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                cur = out.setdefault(m.group(1), {})
                continue
            if cur is None:
                continue
            for key, pat in (("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("vgpr_spill", r"VGPRs Spill: (\d+)"),
                             ("sgpr_spill", r"SGPRs Spill: (\d+)"), ("vgprs", r" VGPRs: (\d+)"), ("agprs", r"AGPRs: (\d+)")):
                m = re.search(pat, line)
                if m:
                    cur[key] = int(m.group(1))
    return out


def check_resources(objs=None, verbose=True):
    if objs is None:
        objs = [os.path.join(OBJ, s + ".o") for s in SOURCES]
    res = kernel_resources(objs)
    problems = []
    for pat, (max_scratch, max_spill) in RESOURCE_BUDGET.items():
        hits = {k: v for k, v in res.items() if pat in k}
        if not hits and res:
            problems.append(f"no kernel matching {pat} in the compiler remarks")
        for k, v in hits.items():
            if v.get("scratch", 0) > max_scratch or v.get("vgpr_spill", 0) > max_spill:
                problems.append(f"{k}: scratch {v.get('scratch')} B/lane (budget {max_scratch}), VGPR spills {v.get('vgpr_spill')} (budget {max_spill})")
            elif verbose:
                print(f"[bornvi] resources ok: {pat}: {v.get('vgprs')} VGPRs + {v.get('agprs')} AGPRs, scratch {v.get('scratch')} B/lane")
    if problems:
        raise RuntimeError("kernel register budget exceeded:\n  " + "\n  ".join(problems))
    return res


if __name__ == "__main__":
    build(force="--force" in sys.argv)
