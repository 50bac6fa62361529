#!/bin/bash
# config 5 only: rocprofv3 kernel stats (+ MFMA counters with a second argument) of scripts/c5_synthetic.py
#   bash scripts/c5_profile.sh tag [pmc]  -> gpurun_out/c5prof_<tag>/{kernel_stats_c5.csv, pmc_mfma_c5.json}
set -u
TAG=${1:-x}
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/c5prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
C5="python3 $REPO/scripts/c5_synthetic.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c5 -- $C5 > $OUT/c5_under_rocprof.txt 2> $OUT/stats_c5.err
cp $(ls $OUT/stats_c5/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_c5.csv
rm -rf $OUT/stats_c5
if [ "${2:-}" = pmc ]; then
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT/pmc_mfma_c5 -- $C5 > /dev/null 2> $OUT/pmc_mfma_c5.err
  python3 $REPO/scripts/pmc_summary.py $OUT/pmc_mfma_c5.json --mfma $OUT/pmc_mfma_c5 > /dev/null
  rm -rf $OUT/pmc_mfma_c5
fi
cut -d, -f1-4 $OUT/kernel_stats_c5.csv | cut -c1-150 | head -24
