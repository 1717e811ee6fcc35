mkdir -p gpurun_out/r4r
timeout 300 ./tests/native/selftest 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_golden.py tests/test_gpu_head_dim_256.py -x -q > gpurun_out/r4r/pytest.log 2>&1; tail -4 gpurun_out/r4r/pytest.log
python tools/small_launch.py 2>&1 | grep -v "^\[" | tee gpurun_out/r4r/small_launch.txt
python tools/small_launch.py --rank 3 2>&1 | grep "^|" | tee -a gpurun_out/r4r/small_launch.txt
python tools/shape_sweep.py 16,512,32,8,128,1 8,1024,32,8,128,1 4,2048,32,8,128,1 2,4096,32,8,128,1 1,8192,32,8,128,1 1,2048,16,8,128,1 1,4096,8,8,128,1 2>&1 | grep "^| " | tee gpurun_out/r4r/sweep.txt
for w in ring_varlen zigzag_varlen llama3; do python bench.py --workload $w --no-cpu-baseline 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:60], d['value'], d['ms_per_step'])"; done | tee gpurun_out/r4r/varlen.txt
