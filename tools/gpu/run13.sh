mkdir -p gpurun_out/r4m
for v in base sf; do echo "== $v"; LD_LIBRARY_PATH=build/variants/$v timeout 300 ./tests/native/selftest 2>&1 | tail -3; RFA_LIB_PATH=build/variants/$v/librfa_hip.so python tools/shape_sweep.py 1,8192,32,8,128,1 1,8192,32,8,128,0 1,16384,32,8,128,1 1,8192,32,32,128,1 2>&1 | grep "^| 1"; done | tee gpurun_out/r4m/sfirst.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_headline.py -x -q 2>&1 | tail -5 | tee gpurun_out/r4m/pytest.txt
