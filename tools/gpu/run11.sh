mkdir -p gpurun_out/r4k
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "split_kv or 128_row or reference_fixture or llama3" > gpurun_out/r4k/pytest.log 2>&1; tail -15 gpurun_out/r4k/pytest.log
python tools/small_launch.py 2>&1 | grep "^|" | tee gpurun_out/r4k/small_launch_split.txt
RFA_FWD_KV_NSPLIT=1 python tools/small_launch.py 2>&1 | grep "^|" | tee gpurun_out/r4k/small_launch_nosplit.txt
python tools/shape_sweep.py 1,2048,16,8,128,1 1,2048,2,1,128,1 8,1024,32,8,128,1 1,8192,32,8,128,1 2>&1 | grep "^| " | tee gpurun_out/r4k/sweep.txt
