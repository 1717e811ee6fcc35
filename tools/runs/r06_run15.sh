mkdir -p gpurun_out/r06b
O=gpurun_out/r06b
SH="8,1024,32,8,128,1 4,2048,32,8,128,1 4,4096,32,8,128,1 8,2048,32,8,128,1 2,8192,32,8,128,1 16,1024,32,8,128,1"
rm -f $O/batch_order_q.txt
for v in base qfast base qfast; do
  if [ $v = base ]; then unset RFA_LIB_PATH; else export RFA_LIB_PATH=build/variants/$v/librfa_hip.so; fi
  echo "== $v" >> $O/batch_order_q.txt
  timeout 300 python tools/shape_sweep.py $SH >> $O/batch_order_q.txt 2>&1
done
unset RFA_LIB_PATH
grep -v amdgpu.ids $O/batch_order_q.txt
rm -f $O/bench_ab.txt
for v in bal ns2 bal ns2; do
  if [ $v = bal ]; then unset RFA_DKDV_WIDE RFA_DKDV_NSPLIT; else export RFA_DKDV_WIDE=1 RFA_DKDV_NSPLIT=2; fi
  timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', round(d['value'],1), round(d['ms_per_step'],4), d['kernels_in_step']['ms'], round(d['power']['avg_w']), round(d['power']['joules_per_step'],3))" >> $O/bench_ab.txt
done
unset RFA_DKDV_WIDE RFA_DKDV_NSPLIT
cat $O/bench_ab.txt
