mkdir -p gpurun_out/r4e
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "128_row or reference_fixture or handoff or smoke or selftest" > gpurun_out/r4e/pytest.log 2>&1; tail -5 gpurun_out/r4e/pytest.log
python tools/shape_sweep.py 8,1024,32,8,128,1 4,2048,32,8,128,1 1,2048,16,8,128,1 1,2048,2,1,128,1 1,4096,32,8,128,1 1,8192,32,8,128,1 2>&1 | grep "^| " | tee gpurun_out/r4e/small.txt
for w in "--workload llama3" "--workload zigzag_varlen" ""; do python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-breakdown $w 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['metric'], round(d['value'],1), round(d['ms_per_step'],4))"; done | tee gpurun_out/r4e/bench.txt
