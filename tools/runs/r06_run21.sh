mkdir -p gpurun_out/r06b
O=gpurun_out/r06b
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "balanced" 2>&1 | tail -3
timeout 300 python tools/shape_sweep.py 1,8192,32,8,64,1 2,4096,32,8,64,1 4,2048,32,8,64,1 1,8192,32,8,128,1 2>&1 | grep "^| [0-9]"
RFA_DKDV_WIDE=1 RFA_DKDV_NSPLIT=2 timeout 300 python tools/shape_sweep.py 1,8192,32,8,64,1 2,4096,32,8,64,1 4,2048,32,8,64,1 2>&1 | grep "^| [0-9]"
