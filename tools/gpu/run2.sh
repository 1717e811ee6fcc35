set -x
mkdir -p gpurun_out/r4b
export RFA_TOL_LOG=$PWD/gpurun_out/r4b/tol.log
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_headline.py "tests/test_gpu_configs.py::test_config3_headline_w8_at_its_stated_shape" tests/test_gpu_golden.py tests/test_gpu_rccl_world1.py -x -q --durations=15 > gpurun_out/r4b/pytest.log 2>&1
tail -30 gpurun_out/r4b/pytest.log
unset RFA_TOL_LOG
python tools/shape_sweep.py 1,8192,32,8,128,1 1,16384,32,8,128,1 1,32768,32,8,128,1 1,16384,32,32,128,1 2>&1 | tee gpurun_out/r4b/sweep_chunked.txt
RFA_DS_SPILL_MAX_BYTES=40000000000 python tools/shape_sweep.py 1,16384,32,8,128,1 1,32768,32,8,128,1 2>&1 | tee gpurun_out/r4b/sweep_unchunked.txt
RFA_BWD_DS_SPILL=0 python tools/shape_sweep.py 1,16384,32,8,128,1 1,32768,32,8,128,1 2>&1 | tee gpurun_out/r4b/sweep_7gemm.txt
cat /sys/class/drm/card0/device/hwmon/hwmon*/freq1_label /sys/class/drm/card0/device/hwmon/hwmon*/freq2_label
(python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-breakdown > gpurun_out/r4b/b.json 2>/dev/null &) ; sleep 14; for i in 1 2 3 4 5 6; do cat /sys/class/drm/card0/device/hwmon/hwmon*/freq1_input /sys/class/drm/card0/device/hwmon/hwmon*/freq2_input /sys/class/drm/card0/device/pp_dpm_sclk 2>/dev/null | tr '\n' ' '; echo; sleep 0.1; done; rocm-smi --showclocks 2>/dev/null | head -20; sleep 5
