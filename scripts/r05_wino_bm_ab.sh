#!/bin/bash
# A/B of the fused Winograd kernel's block shape choice (NBM_WINO_BM=128: always 128-row blocks, the kernel up to round 5; unset: automatic)
# on the detect step, alternating processes on one box.
set -e
mkdir -p gpurun_out/r5l
for i in 1 2 3; do
  NBM_WINO_BM=128 timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-train --no-cpu-baseline --bulk-files 0 --no-split-leg > gpurun_out/r5l/bm128_$i.json 2> gpurun_out/r5l/err.txt
  timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-train --no-cpu-baseline --bulk-files 0 --no-split-leg > gpurun_out/r5l/auto_$i.json 2> gpurun_out/r5l/err.txt
done
for f in gpurun_out/r5l/*.json; do echo -n "$f  "; python scripts/bench_summary.py $f | grep -E "^detect"; done
