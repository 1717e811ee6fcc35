#!/bin/bash
# one gpurun call of the round (scratch script: edited per call)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import sys; sys.path.insert(0,'ring-flash-attention_amd'); from ring_flash_attn import _C; print('build id', _C.load().rfa_build_id().decode())" > $O/build_id.txt 2>&1
( time bash profiles/collect_pmc.sh r05 ) > $O/collect_pmc.log 2>&1
cp $R/gpurun_out/prof/r05_* $O/ 2>/dev/null
B="python bench.py --no-cpu-baseline"
timeout 300 python bench.py > $O/r05_bench_n1_default_flags.json 2> $O/bench_default.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_n1_driver_command.json 2>> $O/bench_default.err
timeout 200 $B --kv-heads 32 > $O/r05_bench_n1_mha.json 2>> $O/bench_default.err
timeout 200 $B --forward-only > $O/r05_bench_n1_forward_only.json 2>> $O/bench_default.err
for wl in ring stripe ring_varlen zigzag_varlen llama3; do timeout 200 $B --workload $wl > $O/r05_bench_n1_$wl.json 2>> $O/bench_default.err; done
{ for vw in 2 4 8; do timeout 200 $B --no-breakdown --virtual-world $vw; done; timeout 200 $B --no-breakdown --virtual-world 8 --exchange ring; } > $O/r05_virtual_ring.txt 2>> $O/bench_default.err
timeout 300 python tools/shape_sweep.py > $O/r05_shape_sweep.md 2>/dev/null
{ timeout 120 python tools/small_launch.py --rank 7 2>/dev/null | grep -v Gloo; timeout 120 python tools/small_launch.py --rank 3 2>/dev/null | grep -v Gloo; } > $O/r05_small_launch_llama3.txt
( timeout 300 ./tests/native/selftest ) > $O/r05_native_selftest.txt 2>&1
( timeout 300 python __graft_entry__.py smoke ) > $O/r05_smoke.txt 2>&1
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=25 ) > $O/r05_pytest_gpu.log 2>&1
tail -8 $O/collect_pmc.log; cut -c1-300 $O/r05_bench_n1_default_flags.json; tail -4 $O/r05_pytest_gpu.log; grep -E "passed|failed" $O/r05_pytest_gpu.log; tail -2 $O/r05_smoke.txt; tail -2 $O/r05_native_selftest.txt; du -sh $R/gpurun_out
