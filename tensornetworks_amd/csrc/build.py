#!/usr/bin/env python3
"""Builds libbornvi_hip.so for gfx950 (MI355X) with hipcc; in-tree output next to the package."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
REPO = os.path.dirname(PKG)
SOURCES = ["api.hip", "kernels_circuit.hip", "kernels_stein.hip", "plan.cpp"]
OUT = os.path.join(PKG, "libbornvi_hip.so")


def build(force=False, verbose=True):
    srcs = [os.path.join(HERE, s) for s in SOURCES]
    deps = srcs + [os.path.join(HERE, h) for h in ("plan.hpp", "kernels.hpp")] + [os.path.join(REPO, "include", "bornvi.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        if verbose:
            print(f"[bornvi] {OUT} is up to date")
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
           "-I" + os.path.join(REPO, "include"), "-I" + HERE, "-o", OUT] + srcs
    if verbose:
        print("[bornvi]", " ".join(cmd))
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
