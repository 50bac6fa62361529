#!/bin/bash
# rocprofv3 kernel stats of scripts/cnot2_timing.py -> gpurun_out/cnot2prof/kernel_stats.csv
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/cnot2prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/scripts/cnot2_timing.py > $OUT/out.txt 2> $OUT/err.txt
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/stats
cut -d, -f1-4 $OUT/kernel_stats.csv | cut -c1-150 | head -24; tail -3 $OUT/out.txt
