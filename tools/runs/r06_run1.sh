mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 300 python -m pytest tests/test_gpu_headline.py -x -q -m gpu > $O/pytest_headline.log 2>&1; echo "pytest rc $?"
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_first.json 2> $O/bench_first.err; echo "bench rc $?"; tail -c 600 $O/bench_first.json
timeout 400 python tools/power_limiters.py --seconds 3 --json $O/power_limiters.json > $O/power_limiters.md 2>&1; echo "limiters rc $?"
timeout 300 python tools/handoff_residency.py > $O/handoff_residency_nt.md 2>&1; echo "handoff rc $?"
RFA_LIB_PATH=build/variants/defpol/librfa_hip.so timeout 300 python tools/handoff_residency.py > $O/handoff_residency_defpol.md 2>&1; echo "handoff defpol rc $?"
for v in base halfreads base halfreads; do
  if [ $v = base ]; then unset RFA_LIB_PATH; else export RFA_LIB_PATH=build/variants/$v/librfa_hip.so; fi
  echo "== $v" >> $O/halfreads_power.txt
  timeout 200 python tools/power_probe.py --seconds 2.5 --phases dkdv,bwd,step >> $O/halfreads_power.txt 2>&1
done
unset RFA_LIB_PATH
cat $O/power_limiters.md | tail -25
