# A/B of the tile-prologue work (fast division, select-only tap masks, single-stage 64-wide tiles) against the library built from the
# previous commit (git archive <commit> birdsoundclassif_amd/csrc include | tar -x -C /tmp/old; make -C /tmp/old/birdsoundclassif_amd/csrc;
# copy the .so to scratch_ab/libnbm_hip_old.so -- selected through NBM_LIB), alternating processes on one box: training step B = 128 and the detect step
mkdir -p gpurun_out/r5p
for i in 1 2 3; do
  NBM_LIB=scratch_ab/libnbm_hip_old.so timeout -k 10 300 python scripts/trainbench.py 128 6 2>&1 | grep "it=" | tail -2 | sed "s/^/old /" | cut -c1-60
  timeout -k 10 300 python scripts/trainbench.py 128 6 2>&1 | grep "it=" | tail -2 | sed "s/^/new /" | cut -c1-60
done | tee gpurun_out/r5p/train_ab.txt
for i in 1 2; do
  NBM_LIB=scratch_ab/libnbm_hip_old.so timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-train --no-cpu-baseline --bulk-files 0 > gpurun_out/r5p/det_old_$i.json 2>/dev/null
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-train --no-cpu-baseline --bulk-files 0 > gpurun_out/r5p/det_new_$i.json 2>/dev/null
done
for f in gpurun_out/r5p/det_*.json; do echo -n "$(basename $f) "; python scripts/bench_summary.py $f | grep -E "^eager|^split" | sed -E "s/.*single \{'ms_per_step': ([0-9.]+).*multi \{'lanes': 2, 'ms_per_step': ([0-9.]+).*/one lane \1  two lanes \2/" | cut -c1-120 | tr '\n' ' '; echo; done | tee gpurun_out/r5p/detect_ab.txt
