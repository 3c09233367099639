mkdir -p gpurun_out/r5f
timeout -k 10 900 python -m pytest tests/test_gpu_lazy.py tests/test_gpu_train.py -x -q 2>&1 | tail -5 &&
for v in 1 0 1 0; do NBM_DIRECT_WGRAD=$v timeout -k 10 300 python scripts/trainbench.py 128 5 2>&1 | grep "it=" | tail -3 | sed "s/^/direct_wgrad=$v /" | cut -c1-110; done | tee gpurun_out/r5f/sink_ab.txt
