set -x
mkdir -p gpurun_out/r4a
export RFA_TOL_LOG=$PWD/gpurun_out/r4a/tol.log
timeout 900 python -m pytest tests/test_gpu_headline.py "tests/test_gpu_configs.py::test_config3_headline_w8_at_its_stated_shape" "tests/test_gpu_rccl_world1.py::test_bench_multi_rank_branches_on_rccl_world_size_1" -x -q --durations=10 > gpurun_out/r4a/pytest.log 2>&1
tail -25 gpurun_out/r4a/pytest.log
unset RFA_TOL_LOG
timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/r4a/bench_base.json 2> gpurun_out/r4a/bench_base.err
cat gpurun_out/r4a/bench_base.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('kernels_in_step',{}).get('ms'), d.get('clock'), d.get('errors'))"
for v in base fwdprio kvprio2; do echo "== $v"; RFA_LIB_PATH=build/variants/$v/librfa_hip.so timeout 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('kernels_in_step',{}).get('ms'))"; done 2>&1 | tee gpurun_out/r4a/variants.txt
for v in base fwdprio kvprio2 base; do echo "== $v"; RFA_LIB_PATH=build/variants/$v/librfa_hip.so timeout 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-breakdown 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done 2>&1 | tee -a gpurun_out/r4a/variants.txt
for w in "--forward-only" "--workload ring" "--workload stripe" "--workload ring_varlen" "--workload ring --forward-only"; do echo "== $w"; timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline $w 2>gpurun_out/r4a/wl.err | tee -a gpurun_out/r4a/workloads.jsonl | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['metric'], d['value'], d['ms_per_step'], d.get('roofline',{}).get('kernel'), d.get('roofline',{}).get('frac'), d.get('errors'))"; done
ls /sys/class/drm/*/device/hwmon/*/ 2>/dev/null | head -30
