#!/bin/bash
# An alternative build of the whole library with extra compiler flags, for A/B timing with scripts/ab_env.py:
#   bash scripts/build_variant.sh NAME "-mllvm -amdgpu-sched-strategy=max-ilp"     -> scripts/ubench/bin/libqgd_NAME.so
#   python3 scripts/ab_env.py QGD_LIB_PATH=scripts/ubench/bin/libqgd_NAME.so
set -e
NAME=$1; shift
FLAGS="$*"
ONLY=${ONLY:-}            # ONLY="qgd_k_sparse qgd_k_grad": the extra flags for these files only
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/quantumgatedesign.jl_amd/csrc
OUT=$ROOT/scripts/ubench/bin
B=$OUT/_build_$NAME
mkdir -p $B
for k in qgd_k_build qgd_k_inverse qgd_k_chain qgd_k_grad qgd_k_sparse qgd_k_forced qgd_k_dense qgd_k_layout qgd_k_tiny; do
  F="$FLAGS"; if [ -n "$ONLY" ] && ! echo " $ONLY " | grep -q " $k "; then F=""; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $F -I$SRC -I$ROOT/include -c $SRC/$k.hip -o $B/$k.o &
done
for k in qgd_host_alloc qgd_host_eval qgd_host_windows qgd_host_comm; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -I$SRC -I$ROOT/include -c $SRC/$k.cpp -o $B/$k.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libqgd_$NAME.so $B/*.o
rm -rf $B
ls -la $OUT/libqgd_$NAME.so
