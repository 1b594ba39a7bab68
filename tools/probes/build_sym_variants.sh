#!/bin/bash
# Builds variants of libbornvi_hip.so that differ only in the symmetric contraction (waves per band) into
# tools/_variants/ (git-ignored, travels to the GPU box).
set -e
cd "$(dirname "$0")/../.."
C=tensornetworks_amd/csrc
O=$C/_obj
mkdir -p tools/_variants
rm -f tools/_variants/libbornvi_sym_*.so
for w in 4 8; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Iinclude -I$C -DBORNVI_SYM_WAVES=$w -c $C/kernels_stein.hip -o tools/_variants/ks_$w.o &
done
wait
for w in 4 8; do
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_variants/libbornvi_sym_waves$w.so \
     $O/api.hip.o $O/kernels_circuit.hip.o $O/kernels_batched.hip.o $O/kernels_adjoint.hip.o $O/plan.cpp.o tools/_variants/ks_$w.o
  rm tools/_variants/ks_$w.o
done
ls tools/_variants/
