mkdir -p gpurun_out/r06b
O=gpurun_out/r06b
timeout 600 python tools/bal_check.py 2,8192,32,8 1,8192,64,8 1,8192,32,16 4,4096,32,8 8,2048,32,8 1,4096,32,8 1,8192,16,4 2,4096,16,4 1,8192,24,8 1,6144,32,8 1,5120,32,8 6,1024,32,8 12,1024,32,8 3,2048,32,8 > $O/bal_check2.txt 2>&1; echo "rc $?"; cat $O/bal_check2.txt | tail -20
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
RFA_DKDV_WIDE=2 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_bal -o run -- python tools/shape_sweep.py 1,8192,32,8,128,1 > $O/prof_bal.log 2>&1
python profiles/summarize_rocpd.py $(find $O/prof_bal -name '*.db' | head -1) > $O/prof_bal.txt 2>&1; rm -rf $O/prof_bal; head -8 $O/prof_bal.txt
timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_bal.json 2> $O/bench_bal.err; echo "bench rc $?"; python - <<'P'
import json
d=json.loads(open('gpurun_out/r06b/bench_bal.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['kernels_in_step']['ms'], d['roofline']['frac'], d['power'])
P
