#!/bin/bash
# rocprofv3 kernel stats of scripts/mid_n_timing.py -> gpurun_out/midn_<tag>/kernel_stats.csv
set -u
TAG=${1:-x}
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/midn_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/scripts/mid_n_timing.py > $OUT/out.txt 2> $OUT/err.txt
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/stats
cut -d, -f1-4 $OUT/kernel_stats.csv | cut -c1-130 | head -30
