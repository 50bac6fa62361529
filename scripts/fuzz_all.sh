#!/bin/bash
# every randomised cross-check of scripts/fuzz_*.py with a fresh seed, one after the other (gpurun -- 'bash scripts/fuzz_all.sh SEED')
S=${1:-5}
mkdir -p gpurun_out
for f in "fuzz_call_sequences.py 60 $S" "fuzz_entry_points.py 24 $S" "fuzz_windows.py 16 $S" "fuzz_small_n.py 60 $S" "fuzz_partitions.py 20 $S"; do
  set -- $f
  echo "== $f" >> gpurun_out/fuzz_all.txt
  timeout -k 10 420 python3 scripts/$1 $2 $3 2>&1 | tail -6 >> gpurun_out/fuzz_all.txt
  rc=${PIPESTATUS[0]}      # (the script's status, not tail's)
  echo "rc=$rc" >> gpurun_out/fuzz_all.txt
  [ "$rc" -ne 0 ] && FAILED=1
done
exit ${FAILED:-0}
