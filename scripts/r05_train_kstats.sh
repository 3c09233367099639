#!/bin/bash
# kernel statistics of the B = 128 training step with / without the composed RPN reader
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  NBM_RPN_COMPOSITE_TRAIN=$v rocprofv3 --kernel-trace --stats --output-format csv -d $O/ts$v -- python3 $R/scripts/trainbench.py 128 3 > $O/train_rocprof_$v.log 2>&1 &&
  find $O/ts$v -name "*kernel_stats.csv" -exec cp {} $O/train_kernel_stats_composite$v.csv \; ; rm -rf $O/ts$v
done
ls -la $O
