#!/bin/bash
# one gpurun call of the round (scratch script: edited per call)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import torch; print(torch.cuda.get_device_name(0))" > $O/device.txt 2>&1
( time timeout 900 python -m pytest tests -m "gpu and not extended" -x -q --durations=20 ) > $O/pytest_core.log 2>&1
( time timeout 900 python -m pytest tests -m gpu -x -q -k "stated_depth" --durations=5 ) > $O/pytest_cfg5_full.log 2>&1
timeout 200 python tools/power_probe.py --seconds 3 > $O/power_probe.json 2> $O/power_probe.err
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
( time timeout 900 bash profiles/collect_scale.sh 1 r05 ) > $O/collect_scale_n1.txt 2>&1
tail -3 $O/pytest_core.log; tail -3 $O/pytest_cfg5_full.log; cat $O/power_probe.json | head -80; cat $O/bench_default.json | cut -c1-600; tail -12 $O/collect_scale_n1.txt
