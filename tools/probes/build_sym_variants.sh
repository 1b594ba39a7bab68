#!/bin/bash
# Builds variants of libbornvi_hip.so that differ only in the symmetric contraction's load shape
# (rows per batch, 1-KiB chunks per row and trip, waves per SIMD) into tools/_variants/ (travels to the GPU box).
set -e
cd "$(dirname "$0")/../.."
C=tensornetworks_amd/csrc
O=$C/_obj
mkdir -p tools/_variants
for v in "8 4 1" "4 4 2" "8 2 2" "2 8 2" "4 4 1" "4 2 2"; do
  set -- $v
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Iinclude -I$C -DBORNVI_SYM_RB=$1 -DBORNVI_SYM_CH=$2 -DBORNVI_SYM_OCC=$3 \
     -c $C/kernels_stein.hip -o tools/_variants/ks_$1_$2_$3.o &
done
wait
for v in "8 4 1" "4 4 2" "8 2 2" "2 8 2" "4 4 1" "4 2 2"; do
  set -- $v
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_variants/libbornvi_sym_$1_$2_$3.so \
     $O/api.hip.o $O/kernels_circuit.hip.o $O/plan.cpp.o tools/_variants/ks_$1_$2_$3.o
  rm tools/_variants/ks_$1_$2_$3.o
done
ls -la tools/_variants/
