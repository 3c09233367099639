mkdir -p gpurun_out/r5g
timeout -k 10 600 python -m pytest tests/test_gpu_split.py tests/test_gpu_ops.py -x -q 2>&1 | tail -5 &&
for v in "0 1" "1 1" "1 0" "0 1" "1 1"; do set -- $v; NBM_SPLIT_BF16=$1 NBM_SPLIT_NN=$2 timeout -k 10 300 python scripts/trainbench.py 128 5 2>&1 | grep "it=" | tail -3 | sed "s/^/split=$1 nn=$2 /" | cut -c1-100; done | tee gpurun_out/r5g/split_train_ab.txt
