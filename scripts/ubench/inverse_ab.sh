#!/bin/bash
# profiles/<round>_inverse_ab.txt: scripts/ubench/bin/inverse_cb_bench (k_inverse_cb beside k_inverse_mfma<64>, one process, interleaved
# rounds) over the batch sizes, the data sets that drive each pivot stage, and the two instantiations forced the other way.
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I quantumgatedesign.jl_amd/csrc -I include scripts/ubench/inverse_cb_bench.hip -o scripts/ubench/bin/inverse_cb_bench
#   gpurun -- 'bash scripts/ubench/inverse_ab.sh'      -> gpurun_out/inverse_ab.txt
O=gpurun_out/inverse_ab.txt; mkdir -p gpurun_out; : > $O
for a in "256 0" "512 0" "550 0" "768 0" "1100 0" "2200 0" "8800 0" "550 1" "550 1 1" "550 3" "550 4" "550 0 0 0" "1100 0 0 1"; do
  echo "== inverse_cb_bench $a" >> $O
  scripts/ubench/bin/inverse_cb_bench $a | grep -v "launch status" >> $O
done
