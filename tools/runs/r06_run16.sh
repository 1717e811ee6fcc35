mkdir -p gpurun_out/r06b
O=gpurun_out/r06b
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "balanced or native_selftest or dkdv_256 or reference_fixture" > $O/pytest_bal.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest_bal.log
timeout 600 python -m pytest tests/test_gpu_headline.py -x -q -m gpu > $O/pytest_headline.log 2>&1; echo "pytest headline rc $?"; tail -3 $O/pytest_headline.log
