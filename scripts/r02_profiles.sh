#!/bin/bash
# Round-2 profile collection on the GPU box (run through gpurun): writes the files that profiles/r02_* are copied from.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/scripts/fwdprofile.py 64 > $O/fwd_layers.txt 2>&1
python3 $R/scripts/fwdprofile.py 64 direct > $O/fwd_layers_direct.txt 2>&1
python3 $R/scripts/trainlayers.py 128 > $O/train_layers.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dense-reference > $O/bench_under_rocprof.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train --no-dense-reference > $O/pmc_$c.log 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_MFMA -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train --no-dense-reference > $O/pmc_MFMA.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_train_$c -- python3 $R/scripts/trainbench.py 128 2 > $O/pmc_train_$c.log 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_train_MFMA -- python3 $R/scripts/trainbench.py 128 2 > $O/pmc_train_MFMA.log 2>&1
# plain bench line (no profiler attached): the numbers DESIGN.md quotes
python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
GF=$(python3 -c "import json,sys; print(json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])['roofline']['executed_GFLOP_per_launch'])")
# dominant kernels: forward = wino23_fused_kernel, mean over ALL its launches of a step (pairs with its average duration in
# bench_kernel_stats.csv); backward = the weight-gradient kernel's largest launch
python3 $R/scripts/pmc_dominant.py $O/pmc_dominant.json "wino23_fused_kernel" $GF $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_MFMA "wino23_fused_kernel<128,64>, mean over all launches of the B = 64 detect step (finest FPN map on demand)" all > /dev/null
python3 $R/scripts/pmc_dominant.py $O/pmc_rows.json "wino23_rows_tiles_kernel" 0 $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_MFMA "row half of the Winograd input transform, listed tiles of fpn.out_convs.4, 64 images" > /dev/null
python3 $R/scripts/pmc_dominant.py $O/pmc_wgrad.json "igemm_tn_kernel<128, 0, 0>" 0 $O/pmc_train_FETCH_SIZE $O/pmc_train_WRITE_SIZE $O/pmc_train_MFMA "largest weight-gradient launch of the B = 128 training step" > /dev/null
# per-kernel means of the counters (small text files; the raw csv stays on the box)
for d in pmc_FETCH_SIZE pmc_WRITE_SIZE pmc_MFMA pmc_train_FETCH_SIZE pmc_train_WRITE_SIZE pmc_train_MFMA; do
  python3 $R/scripts/pmc_summarize.py $O/$d > $O/$d.summary.txt
  rm -rf $O/$d
done
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
rm -rf $O/stats
ls -la $O
