# A/B of the half-step forward kernel (NBM_H16=3; NBM_H16_MIN_NK: smallest K / 32 that takes it) on the detect step and the training step,
# alternating processes on one box
mkdir -p gpurun_out/r5r
for cfg in "0 9" "3 9" "3 1" "0 9" "3 9" "3 1"; do
  set -- $cfg
  NBM_H16=$1 NBM_H16_MIN_NK=$2 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-train --no-cpu-baseline --bulk-files 0 --no-split-leg > gpurun_out/r5r/det_$1_$2_$RANDOM.json 2>/dev/null
done
for f in gpurun_out/r5r/det_*.json; do echo -n "$(basename $f) "; python scripts/bench_summary.py $f | grep -E "^eager" | sed -E "s/.*single \{'ms_per_step': ([0-9.]+).*multi \{'lanes': 2, 'ms_per_step': ([0-9.]+).*/one lane \1  two lanes \2/" | cut -c1-100; done | tee gpurun_out/r5r/detect_ab.txt
for cfg in "0 9" "3 9" "3 1" "0 9" "3 9" "3 1"; do
  set -- $cfg
  NBM_H16=$1 NBM_H16_MIN_NK=$2 timeout -k 10 300 python scripts/trainbench.py 128 6 2>&1 | grep "it=" | tail -2 | sed "s/^/h16=$1 min_nk=$2 /" | cut -c1-75
done | tee gpurun_out/r5r/train_ab.txt
