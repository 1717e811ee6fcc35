mkdir -p gpurun_out/r06b
O=gpurun_out/r06b
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu_full.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest_gpu_full.log
timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'],1), round(d['ms_per_step'],4), d['kernels_in_step']['ms'], round(d['power']['avg_w']), d['roofline']['frac'])"
