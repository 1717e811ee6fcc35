mkdir -p gpurun_out/r06b
O=gpurun_out/r06b
rm -f $O/bench_ab3.txt
for v in bal ns2 bal ns2 bal ns2; do
  if [ $v = bal ]; then unset RFA_DKDV_WIDE RFA_DKDV_NSPLIT; else export RFA_DKDV_WIDE=1 RFA_DKDV_NSPLIT=2; fi
  timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', round(d['value'],1), round(d['ms_per_step'],4), d['kernels_in_step']['ms'])" >> $O/bench_ab3.txt
done
unset RFA_DKDV_WIDE RFA_DKDV_NSPLIT
cat $O/bench_ab3.txt
timeout 300 python tools/bal_check.py 1,8192,32,8 4,2048,32,8 2,4096,32,8 2>&1 | grep "^| [0-9]"
