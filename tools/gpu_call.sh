#!/bin/bash
# one gpurun call of the round (scratch script: edited per call)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 1200 python -m pytest tests -m gpu -x -q -k "config2 or config4 or config5 or rccl or torch_compile or forced_256" --durations=20 ) > $O/pytest_changed.log 2>&1
timeout 120 build/tools/mfma_power_probe 2.5 > $O/mfma_power_probe.txt 2>&1
timeout 200 python tools/power_probe.py --seconds 2.5 > $O/power_probe.json 2> $O/power_probe.err
( time timeout 900 bash profiles/collect_scale.sh 1 r05 ) > $O/collect_scale_n1.txt 2>&1
grep -E "passed|failed|error|Error|s call|s setup" $O/pytest_changed.log | tail -30; cat $O/mfma_power_probe.txt; tail -12 $O/collect_scale_n1.txt; du -sh $R/gpurun_out
