mkdir -p gpurun_out/r5f
timeout -k 10 300 python scripts/small_launches.py 128 2>&1 | grep -v amdgpu.ids > gpurun_out/r5f/small_launches.txt
NBM_SHORTK_MAX=8 timeout -k 10 300 python scripts/fwdprofile.py 64 2>&1 | grep -v amdgpu.ids > gpurun_out/r5f/fwd_layers_shortk8.txt
NBM_SHORTK_MAX=7 timeout -k 10 300 python scripts/fwdprofile.py 64 2>&1 | grep -v amdgpu.ids > gpurun_out/r5f/fwd_layers_shortk7.txt
head -50 gpurun_out/r5f/small_launches.txt
