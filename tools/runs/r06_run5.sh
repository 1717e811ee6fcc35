O=gpurun_out/r06
mkdir -p $O
timeout 900 python tools/plan_sweep.py --dump $O/plan_sweep_dump_after2.json > $O/plan_sweep_after2.md 2>&1; echo "plan sweep rc $?"; grep -v amdgpu.ids $O/plan_sweep_after2.md | tail -16
timeout 300 python -m pytest tests/test_gpu_plan_rules.py -x -q -m gpu 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py -x -q -m gpu > $O/pytest_kernels_configs.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_kernels_configs.log
for vw in 2 8; do timeout 200 python bench.py --no-cpu-baseline --no-breakdown --virtual-world $vw | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['virtual_ring'])"; done
