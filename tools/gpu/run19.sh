mkdir -p gpurun_out/r4s
timeout 900 python -m pytest tests/test_gpu_head_dim_256.py -x -q 2>&1 | tail -3 | tee gpurun_out/r4s/pytest.txt
python tools/shape_sweep.py 1,8192,20,5,192,1 1,8192,20,5,192,0 1,8192,24,8,160,1 1,8192,16,4,256,1 1,8192,32,8,128,1 2>&1 | grep "^| " | tee gpurun_out/r4s/sweep.txt
