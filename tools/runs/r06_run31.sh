mkdir -p gpurun_out/r06b
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "hip_graph" > gpurun_out/r06b/graph_test.log 2>&1; echo rc $?; grep -v "Extension modules" gpurun_out/r06b/graph_test.log | tail -40
