mkdir -p gpurun_out/r4u
bash profiles/collect_pmc.sh r04 > gpurun_out/r4u/collect.log 2>&1; tail -12 gpurun_out/r4u/collect.log
cp profiles/r04_traffic.json gpurun_out/prof/ 2>/dev/null
python bench.py > gpurun_out/r4u/bench_default.json 2> gpurun_out/r4u/bench_default.err; python -c "
import json; d=json.load(open('gpurun_out/r4u/bench_default.json')); print(d['value'], d['ms_per_step'], d['kernels_in_step']['ms'], d['roofline']['frac'], d['roofline'].get('traffic_over_algorithmic'), d['roofline'].get('profiled'), d['cpu_baseline']['value'], d.get('errors'))"
python bench.py --kv-heads 32 --no-cpu-baseline > gpurun_out/r4u/bench_mha.json 2>/dev/null
python bench.py --forward-only --no-cpu-baseline > gpurun_out/r4u/bench_fwd_only.json 2>/dev/null
for w in ring stripe ring_varlen zigzag_varlen llama3; do python bench.py --workload $w --no-cpu-baseline > gpurun_out/r4u/bench_$w.json 2>/dev/null; done
for f in gpurun_out/r4u/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', round(d['value'],1), d.get('roofline',{}).get('kernel'), round(d.get('roofline',{}).get('frac',0),3), d.get('roofline',{}).get('traffic_over_algorithmic'))"; done
python tools/shape_sweep.py > gpurun_out/r4u/shape_sweep.md 2>&1; tail -22 gpurun_out/r4u/shape_sweep.md
mkdir -p gpurun_out/r4u
export RFA_TOL_LOG=$PWD/gpurun_out/r4u/tol.log
timeout 1700 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/r4u/pytest_gpu.log 2>&1; tail -25 gpurun_out/r4u/pytest_gpu.log
unset RFA_TOL_LOG
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
