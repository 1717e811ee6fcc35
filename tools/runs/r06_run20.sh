mkdir -p gpurun_out/r06b
O=gpurun_out/r06b
timeout 1500 python tools/plan_sweep.py --dump $O/plan_sweep_dump_c.json > $O/plan_sweep_c.md 2>&1; echo "sweep rc $?"; tail -12 $O/plan_sweep_c.md
timeout 1800 python -m pytest tests -q -m gpu > $O/pytest_gpu_full2.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest_gpu_full2.log
