O=gpurun_out/inverse_ab.txt; : > $O
for a in "256 0" "512 0" "550 0" "768 0" "1100 0" "2200 0" "8800 0" "550 1" "550 1 1" "550 3" "550 4" "550 0 0 0" "1100 0 0 1"; do echo "== inverse_cb_bench $a" >> $O; scripts/ubench/bin/inverse_cb_bench $a | grep -v "launch status" >> $O; done
