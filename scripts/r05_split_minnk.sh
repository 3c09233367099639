#!/bin/bash
# A/B of the split route's K threshold (NBM_SPLIT_MIN_NK: K32 steps; 9 = library default) on the detect step's split_bf16 leg.
set -e
mkdir -p gpurun_out/r5j
for nk in 9 8 9 8; do
  NBM_SPLIT_MIN_NK=$nk timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-train --no-cpu-baseline --bulk-files 0 > gpurun_out/r5j/bench_nk${nk}_$RANDOM.json 2> gpurun_out/r5j/err.txt
done
for f in gpurun_out/r5j/bench_nk*.json; do echo $f; python scripts/bench_summary.py $f | grep -E "^detect|^split"; done
