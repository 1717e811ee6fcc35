mkdir -p gpurun_out/r06b
O=gpurun_out/r06b
rm -f $O/bench_ab2.txt
for v in prev new prev new; do
  if [ $v = new ]; then unset RFA_LIB_PATH; else export RFA_LIB_PATH=build/variants/$v/librfa_hip.so; fi
  timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', round(d['value'],1), round(d['ms_per_step'],4), d['kernels_in_step']['ms'], round(d['power']['avg_w']), round(d['power']['joules_per_step'],3))" >> $O/bench_ab2.txt
done
unset RFA_LIB_PATH
cat $O/bench_ab2.txt
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "balanced or reference_fixture" 2>&1 | tail -3
