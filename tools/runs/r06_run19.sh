mkdir -p gpurun_out/r06b
O=gpurun_out/r06b
timeout 1500 python tools/plan_sweep.py --dump $O/plan_sweep_dump_b.json > $O/plan_sweep_b.md 2>&1; echo "rc $?"; tail -60 $O/plan_sweep_b.md
