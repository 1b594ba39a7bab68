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
SOURCES = ["api.hip", "kernels_circuit.hip", "kernels_stein.hip", "kernels_batched.hip", "kernels_adjoint.hip", "plan.cpp"]
HEADERS = [os.path.join(HERE, h) for h in ("plan.hpp", "kernels.hpp")] + [os.path.join(REPO, "include", "bornvi.h")]
OUT = os.path.join(PKG, "libbornvi_hip.so")
OBJ = os.path.join(HERE, "_obj")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-I" + os.path.join(REPO, "include"), "-I" + HERE]


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
            print("[bornvi]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    run([hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", OUT] + objs)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
