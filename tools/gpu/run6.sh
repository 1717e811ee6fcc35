mkdir -p gpurun_out/r4f
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_head_dim_256.py -x -q > gpurun_out/r4f/pytest.log 2>&1; tail -5 gpurun_out/r4f/pytest.log
for w in "--workload llama3" "--workload zigzag_varlen" "--workload ring_varlen" ""; do python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-breakdown $w 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['metric'], round(d['value'],1), round(d['ms_per_step'],4))"; done | tee gpurun_out/r4f/bench.txt
python - <<'PY'
import torch, sys
sys.path.insert(0, 'ring-flash-attention_amd')
print("max mem allocated MB", torch.cuda.max_memory_allocated() / 1e6)
PY
