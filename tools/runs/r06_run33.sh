mkdir -p gpurun_out/r06b
O=gpurun_out/r06b
rm -f $O/knobs.txt
for v in base ka2 ka4 base kb1 kb3 base prio2 prio4 base; do
  if [ $v = base ]; then unset RFA_LIB_PATH; else export RFA_LIB_PATH=build/variants/$v/librfa_hip.so; fi
  timeout 300 python bench.py --steps 250 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', round(d['value'],1), round(d['ms_per_step'],4), d['kernels_in_step']['ms'])" >> $O/knobs.txt
done
cat $O/knobs.txt
