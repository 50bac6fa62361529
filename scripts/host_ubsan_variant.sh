#!/bin/bash
# libqgd_hip.so with its HOST side (qgd_host_*.cpp) under UndefinedBehaviorSanitizer (clang ignores the option for the device code):
#   bash scripts/host_ubsan_variant.sh   -> scripts/ubench/bin/libqgd_ubsan.so
#   QGD_LIB_PATH=scripts/ubench/bin/libqgd_ubsan.so UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 python -m pytest tests -m gpu   (on the GPU box)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/quantumgatedesign.jl_amd/csrc
OUT=$ROOT/scripts/ubench/bin
B=$OUT/_build_ubsan
mkdir -p $B
cp $SRC/_build/qgd_k_*.o $B/
for k in qgd_host_alloc qgd_host_eval qgd_host_windows qgd_host_comm; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -Wno-unused-function -Wno-option-ignored -fsanitize=undefined -fno-sanitize=vptr -I$SRC -I$ROOT/include -c $SRC/$k.cpp -o $B/$k.o &
done
wait
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.ubsan_standalone-x86_64.so | head -1)      # the runtime as a shared object beside the library
cp $RT $OUT/
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libqgd_ubsan.so $B/*.o $OUT/$(basename $RT) -Wl,-rpath,'$ORIGIN' 
rm -rf $B
ls -la $OUT/libqgd_ubsan.so
