mkdir -p gpurun_out/r06b
timeout 900 python tools/bal_soak.py 600 > gpurun_out/r06b/bal_soak.txt 2>&1; echo rc $?; grep -v amdgpu.ids gpurun_out/r06b/bal_soak.txt
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "hip_graph" 2>&1 | tail -2
