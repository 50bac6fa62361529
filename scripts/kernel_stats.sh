#!/bin/bash
# rocprofv3 per-kernel statistics of the cnot3 headline evaluation (bench.py's timed loop, nothing else):
#   gpurun -- bash scripts/kernel_stats.sh <tag> [QGD_PATHS value]   -> gpurun_out/kstats_<tag>.csv
set -u
TAG=${1:-x}; export QGD_PATHS=${2:-}
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/kstats_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-large-n --no-with-history --no-cnot2 > $OUT/bench.json 2> $OUT/err.txt
cp $(ls $OUT/*/*kernel_stats.csv | head -1) $REPO/gpurun_out/kstats_$TAG.csv
rm -rf $OUT/*/
cut -d, -f1-4 $REPO/gpurun_out/kstats_$TAG.csv | head -24
