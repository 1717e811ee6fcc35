mkdir -p gpurun_out/r4l
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_golden.py tests/test_gpu_flash_attn_shim.py tests/test_gpu_head_dim_256.py -x -q > gpurun_out/r4l/pytest.log 2>&1; tail -6 gpurun_out/r4l/pytest.log
python tools/small_launch.py 2>&1 | grep -v "^\[" | tee gpurun_out/r4l/small_launch.txt
python tools/small_launch.py --rank 3 2>&1 | grep "^|" | tee -a gpurun_out/r4l/small_launch.txt
python tools/shape_sweep.py 1,2048,16,8,128,1 1,2048,2,1,128,1 2>&1 | grep "^| " | tee gpurun_out/r4l/sweep.txt
