#!/bin/bash
# Regenerates the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo root):
#   bash scripts/collect_profiles.sh r05
# -> gpurun_out/prof_<tag>/{kernel_stats_cnot3.csv, kernel_stats_c5.csv, bench_under_rocprof.json, pmc_*.json};
# copy what is to be judged into profiles/.  Counters are collected in their own passes (--pmc with --kernel-trace only;
# FETCH_SIZE and WRITE_SIZE cannot share a pass), exactly as MI355X_MICROARCH.md 'HBM' / 'rocprofv3 PMC slots' prescribe.
set -u
TAG=${1:-r05}
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-large-n --no-with-history --no-cnot2"
BENCH_S="python3 $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-large-n --no-with-history --no-cnot2"
C5="python3 $REPO/scripts/c5_synthetic.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cnot3 -- $BENCH > $OUT/bench_under_rocprof.json 2> $OUT/stats_cnot3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c5 -- $C5 > $OUT/c5_under_rocprof.txt 2> $OUT/stats_c5.err
cp $(ls $OUT/stats_cnot3/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_cnot3.csv
cp $(ls $OUT/stats_c5/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_c5.csv
for W in cnot3 c5; do
  if [ $W = cnot3 ]; then CMD=$BENCH_S; else CMD=$C5; fi
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$W -- $CMD > /dev/null 2> $OUT/pmc_fetch_$W.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$W -- $CMD > /dev/null 2> $OUT/pmc_write_$W.err
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT/pmc_mfma_$W -- $CMD > /dev/null 2> $OUT/pmc_mfma_$W.err
  python3 $REPO/scripts/pmc_summary.py $OUT/pmc_fetch_write_$W.json FETCH_SIZE=$OUT/pmc_fetch_$W WRITE_SIZE=$OUT/pmc_write_$W > /dev/null
  python3 $REPO/scripts/pmc_summary.py $OUT/pmc_mfma_$W.json --mfma $OUT/pmc_mfma_$W > /dev/null
done
# keep the merged directory small: the raw per-dispatch CSVs stay on the box
for W in cnot3 c5; do rm -rf $OUT/stats_$W $OUT/pmc_fetch_$W $OUT/pmc_write_$W $OUT/pmc_mfma_$W; done
ls -la $OUT


cd $REPO && python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
tail -c 400 $OUT/bench.json
