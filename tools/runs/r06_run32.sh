mkdir -p gpurun_out/r06b
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "hip_graph or balanced" > gpurun_out/r06b/graph_test.log 2>&1; echo rc $?; grep -v "Extension modules\|File \"/usr" gpurun_out/r06b/graph_test.log | tail -8
timeout 300 python tools/bal_soak.py 200 2>&1 | grep -v amdgpu.ids
