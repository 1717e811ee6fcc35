mkdir -p gpurun_out/r06b
O=gpurun_out/r06b
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "persistent or fwd_ or reference_fixture or hip_graph or smoke" > $O/pytest_fwd.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest_fwd.log
timeout 300 python tools/fwd_persist_check.py > $O/fwd_persist_final.txt 2>&1; grep "^| " $O/fwd_persist_final.txt
