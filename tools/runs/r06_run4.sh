O=gpurun_out/r06
mkdir -p $O
timeout 900 python tools/plan_sweep.py --dump $O/plan_sweep_dump_after.json > $O/plan_sweep_after.md 2>&1; echo "plan sweep rc $?"; grep -v amdgpu.ids $O/plan_sweep_after.md | tail -40
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_headline.py -x -q -m gpu > $O/pytest_kernels.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_kernels.log
timeout 200 python tools/small_launch.py --rank 7 2>/dev/null | grep -v Gloo
timeout 200 python tools/small_launch.py --rank 3 2>/dev/null | grep -v Gloo
timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline | cut -c1-300
