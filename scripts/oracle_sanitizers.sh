#!/bin/bash
# The CPU oracle (test infrastructure, oracle/qgd_oracle.c) under AddressSanitizer + UndefinedBehaviorSanitizer, driven by its own
# CPU test file (GPU sanitizers are not available on the pool; this is the CPU-side run the environment allows):
#   bash scripts/oracle_sanitizers.sh        -> prints the pytest tail and any sanitizer report
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
gcc -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fopenmp -fPIC -std=c99 -shared -o $T/libqgd_oracle.so $ROOT/oracle/qgd_oracle.c -lm
cd $ROOT
QGD_ORACLE_LIB=$T/libqgd_oracle.so LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
  ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python3 -m pytest tests/test_oracle.py -x -q -m "not gpu" 2>&1 | tail -15
rm -rf $T
