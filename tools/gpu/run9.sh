mkdir -p gpurun_out/r4i
export RFA_TOL_LOG=$PWD/gpurun_out/r4i/tol.log
timeout 1700 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/r4i/pytest_gpu.log 2>&1; tail -25 gpurun_out/r4i/pytest_gpu.log
unset RFA_TOL_LOG
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
