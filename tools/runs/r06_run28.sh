# final build: PMC passes + traffic file + kernel trace, the bench lines that quote them, the GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
python -c "import sys; sys.path.insert(0,'ring-flash-attention_amd'); from ring_flash_attn import _C; print('build id', _C.load().rfa_build_id().decode())" > $O/build_id.txt 2>&1
bash profiles/collect_pmc.sh r06 > $O/collect_pmc.log 2>&1
cp $R/gpurun_out/prof/r06_* $O/ 2>/dev/null
timeout 300 python bench.py > $O/r06_bench_n1_default_flags.json 2> $O/bench.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_n1_driver_command.json 2>> $O/bench.err
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 ) > $O/r06_pytest_gpu.log 2>&1
tail -4 $O/r06_pytest_gpu.log; cut -c1-220 $O/r06_bench_n1_default_flags.json; cut -c1-220 $O/r06_bench_n1_driver_command.json; cat $O/build_id.txt
