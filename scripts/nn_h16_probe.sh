# half-step data-gradient kernel (NBM_NN_H16, default on) against the two-stage kernel: launch by launch (scripts/dgrad_ablate.py, shipped
# library, first column) and the B = 128 training step, alternating processes on one box
mkdir -p gpurun_out/r5t
for v in 0 1; do echo "NBM_NN_H16=$v"; NBM_NN_H16=$v timeout -k 10 200 python scripts/dgrad_ablate.py 128 10 2>&1 | grep "TF/s" | cut -c1-100; done | tee gpurun_out/r5t/launches.txt
for v in 0 1 0 1 0 1; do NBM_NN_H16=$v timeout -k 10 300 python scripts/trainbench.py 128 6 2>&1 | grep "it=" | tail -2 | sed "s/^/nn_h16=$v /" | cut -c1-70; done | tee gpurun_out/r5t/train_ab.txt
