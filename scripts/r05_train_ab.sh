mkdir -p gpurun_out/r5d
timeout -k 10 600 python -m pytest tests/test_gpu_lazy.py -x -q -k "composed_with_the_output_convolution_in_training" 2>&1 | tail -3 &&
for v in 1 0 1 0; do NBM_RPN_COMPOSITE_TRAIN=$v timeout -k 10 300 python scripts/trainbench.py 128 5 2>&1 | grep "it=" | tail -4 | sed "s/^/composite=$v /" | cut -c1-160; done | tee gpurun_out/r5d/train_ab.txt
