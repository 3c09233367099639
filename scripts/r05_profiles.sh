#!/bin/bash
# Round-5 profile collection on the GPU box (run through gpurun): writes the files that profiles/r05_* are copied from.
# Parts (argument, default all): layers | stats | pmc_detect | pmc_train | bench
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PART=${1:-all}
DET="--steps 6 --warmup 2 --no-cpu-baseline --no-dense-reference --bulk-files 0 --lanes 1 --no-split-leg"
if [ $PART = all ] || [ $PART = layers ]; then
  python3 $R/scripts/fwdprofile.py 64 > $O/fwd_layers.txt 2>&1
  python3 $R/scripts/trainlayers.py 128 > $O/train_layers.txt 2>&1
  NBM_SPLIT_BF16=1 python3 $R/scripts/fwdprofile.py 64 > $O/fwd_layers_split.txt 2>&1
fi
if [ $PART = all ] || [ $PART = stats ]; then
  # the detect leg alone (the training forward launches the same kernel templates at B = 128: their durations must not mix into the
  # average that is compared with bench.py's live figure for the dominant kernel)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/dstats -- python3 $R/bench.py $DET --no-train > $O/detect_under_rocprof.log 2>&1
  find $O/dstats -name "*kernel_stats.csv" -exec cp {} $O/detect_kernel_stats.csv \;
  rm -rf $O/dstats
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/tstats -- python3 $R/scripts/trainbench.py 128 3 > $O/train_under_rocprof.log 2>&1
  find $O/tstats -name "*kernel_stats.csv" -exec cp {} $O/train_kernel_stats.csv \;
  rm -rf $O/tstats
fi
if [ $PART = all ] || [ $PART = pmc_detect ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 2 --warmup 1 --lanes 1 --no-cpu-baseline --no-train --no-dense-reference --bulk-files 0 > $O/pmc_$c.log 2>&1
  done
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_MFMA -- python3 $R/bench.py --steps 2 --warmup 1 --lanes 1 --no-cpu-baseline --no-train --no-dense-reference --bulk-files 0 > $O/pmc_MFMA.log 2>&1
fi
if [ $PART = all ] || [ $PART = pmc_train ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/pmct_$c -- python3 $R/scripts/trainbench.py 128 2 > $O/pmct_$c.log 2>&1
  done
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmct_MFMA -- python3 $R/scripts/trainbench.py 128 2 > $O/pmct_MFMA.log 2>&1
  python3 $R/scripts/pmc_dominant.py $O/pmc_wgrad.json "igemm_tn_kernel" 0 $O/pmct_FETCH_SIZE $O/pmct_WRITE_SIZE $O/pmct_MFMA "igemm_tn_kernel<...> (weight gradients), mean over all launches of B = 128 training steps (scripts/trainbench.py)" all > /dev/null
  python3 $R/scripts/pmc_dominant.py $O/pmc_dgrad.json "igemm_nn_kernel" 0 $O/pmct_FETCH_SIZE $O/pmct_WRITE_SIZE $O/pmct_MFMA "igemm_nn_kernel<...> (data gradients), mean over all launches of B = 128 training steps (scripts/trainbench.py)" all > /dev/null
  for d in pmct_FETCH_SIZE pmct_WRITE_SIZE pmct_MFMA; do
    python3 $R/scripts/pmc_summarize.py $O/$d > $O/$d.summary.txt
    rm -rf $O/$d
  done
fi
if [ $PART = all ] || [ $PART = bench ]; then
  # plain bench line (no profiler attached): the numbers DESIGN.md quotes
  python3 $R/bench.py --steps 20 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err
fi
if [ $PART = all ] || [ $PART = pmc_detect ]; then
  GF=$(python3 -c "import json,sys; print(json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])['roofline']['executed_GFLOP_per_launch'])" 2>/dev/null || echo 0)
  python3 $R/scripts/pmc_dominant.py $O/pmc_dominant.json "igemm_h16_kernel" $GF $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_MFMA "igemm_h16_kernel<3> (the deep-K kernel; half-step form of igemm_kernel<128,128,64,64,A_FAST,EPI_STD,2>), mean over all launches of the B = 64 detect step (FPN levels 0 and 1 on demand)" all > /dev/null
  python3 $R/scripts/pmc_dominant.py $O/pmc_split.json "igemm_split_kernel" 0 $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_MFMA "igemm_split_kernel<REM> (opt-in NBM_SPLIT_BF16=1: the split_bf16 leg of the same command), mean over all its launches of B = 64 detect steps" all > /dev/null
  python3 $R/scripts/pmc_dominant.py $O/pmc_fused.json "wino23_fused_kernel" 0 $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_MFMA "wino23_fused_kernel (both block shapes: 128 x 128 and 96 x 128), mean over all launches of the B = 64 detect step" all > /dev/null
  for d in pmc_FETCH_SIZE pmc_WRITE_SIZE pmc_MFMA; do
    python3 $R/scripts/pmc_summarize.py $O/$d > $O/$d.summary.txt
    rm -rf $O/$d
  done
fi
ls -la $O
