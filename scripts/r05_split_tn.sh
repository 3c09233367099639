mkdir -p gpurun_out/r5g
timeout -k 10 300 python scripts/split_tn_probe.py 2>&1 | grep -v amdgpu | tee gpurun_out/r5g/split_tn_probe.txt
for v in "1 1 1" "1 1 0" "0 1 1" "1 1 1"; do set -- $v; NBM_SPLIT_BF16=$1 NBM_SPLIT_NN=$2 NBM_SPLIT_TN=$3 timeout -k 10 300 python scripts/trainbench.py 128 5 2>&1 | grep "it=" | tail -3 | sed "s/^/split=$1 nn=$2 tn=$3 /" | cut -c1-105; done | tee gpurun_out/r5g/split_train_ab2.txt
