#!/bin/bash
# instruction-cache counters of the inverse kernels (scripts/ubench/inverse_cb_bench)
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
rocprofv3 -L > $O/counters_avail.txt 2>&1
grep -o -E "\b(SQC_[A-Z_]*ICACHE[A-Z_]*|SQ_IFETCH[A-Z_]*|SQ_WAIT_INST[A-Z_]*|SQC_INST[A-Z_]*)\b" $O/counters_avail.txt | sort -u > $O/counters_icache.txt
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace -d $O/icache1 -o ic -f csv -- $GRAFT_REPO_ROOT/scripts/ubench/bin/inverse_cb_bench 550 0 > $O/icache1.log 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $O/icache2 -o ic -f csv -- $GRAFT_REPO_ROOT/scripts/ubench/bin/inverse_cb_bench 550 0 > $O/icache2.log 2>&1
echo done
