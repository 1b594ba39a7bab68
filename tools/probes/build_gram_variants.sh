#!/bin/bash
# Timing-only variants of the Gram builder (compile-time switches of kernels_stein.hip) into tools/_variants/
# (travels to the GPU box).   tools/probes/build_gram_variants.sh "NAME=-DFLAG=1 ..." ...
set -e
cd "$(dirname "$0")/../.."
C=tensornetworks_amd/csrc
O=$C/_obj
mkdir -p tools/_variants
rm -f tools/_variants/libbornvi_gram_*.so
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Iinclude -I$C $flags -c $C/kernels_stein.hip -o tools/_variants/ks_$name.o &
done
wait
for spec in "$@"; do
  name=${spec%%=*}
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_variants/libbornvi_gram_$name.so \
     $O/api.hip.o $O/kernels_circuit.hip.o $O/kernels_circuit8.hip.o $O/kernels_batched.hip.o $O/kernels_adjoint.hip.o $O/plan.cpp.o tools/_variants/ks_$name.o
  rm tools/_variants/ks_$name.o
done
ls tools/_variants/
