#!/bin/bash
# Builds variants of libbornvi_hip.so that differ only in the symmetric contraction's load scheme into tools/_variants/
# (git-ignored, travels to the GPU box): "WIN OCC" = rolling window of WIN loads (0: batched 8 rows x 4 chunks), OCC waves/SIMD.
set -e
cd "$(dirname "$0")/../.."
C=tensornetworks_amd/csrc
O=$C/_obj
mkdir -p tools/_variants
rm -f tools/_variants/libbornvi_sym_*.so
VARS=("0 1 0" "0 1 2" "0 2 0" "0 2 2")
for v in "${VARS[@]}"; do
  set -- $v
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Iinclude -I$C -DBORNVI_SYM_WIN=$1 -DBORNVI_SYM_OCC=$2 -DBORNVI_SYM_ABLATE=$3 \
     -c $C/kernels_stein.hip -o tools/_variants/ks_$1_$2_$3.o &
done
wait
for v in "${VARS[@]}"; do
  set -- $v
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_variants/libbornvi_sym_w$1_o$2_a$3.so \
     $O/api.hip.o $O/kernels_circuit.hip.o $O/plan.cpp.o tools/_variants/ks_$1_$2_$3.o
  rm tools/_variants/ks_$1_$2_$3.o
done
ls tools/_variants/
