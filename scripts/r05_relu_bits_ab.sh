# A/B of the ReLU masks as bits (NBM_RELU_BITS, functional.relu_bits_note / nbm_gemm_desc.bits_out / nbm_bwd_desc.mask_bits), B = 128 training step,
# alternating processes on one box
mkdir -p gpurun_out/r5v
for v in 0 1 0 1 0 1; do NBM_RELU_BITS=$v timeout -k 10 300 python scripts/trainbench.py 128 6 2>&1 | grep "it=" | tail -2 | sed "s/^/relu_bits=$v /" | cut -c1-72; done | tee gpurun_out/r5v/train_ab.txt
