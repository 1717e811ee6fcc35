#!/bin/bash
# one gpurun call of the round (scratch script: edited per call)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
{
echo "# The dK/dV kernel of the headline launch with parts of its loop compiled out (RFA_KV_X_* / RFA_SPILL_PROBE measurement"
echo "# macros: results invalid, timing only), each looped for 3 s under tools/power_probe.py: ms per launch, board watts, joules."
echo "# If the kernel is energy-bound at the 1400 W cap, every variant that stays at the cap gets faster by exactly the joules it drops."
echo "| variant | ms | TFLOP/s | avg W | max W | J per launch |"
echo "|---|---|---|---|---|---|"
for v in base xspill xsync xvalu xlds xload xall base; do
  if [ $v = base ]; then unset RFA_LIB_PATH; else export RFA_LIB_PATH=$R/build/variants/$v/librfa_hip.so; fi
  timeout 100 python tools/power_probe.py --seconds 3 --phases dkdv 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)['dkdv']
print('| $v | %.4f | %.0f | %.0f | %.0f | %.3f |' % (d['ms_per_iter'], d['tflops'], d['avg_w'], d['max_w'], d['joules_per_iter']))"
done
unset RFA_LIB_PATH
} > $O/r05_dkdv_energy_decomposition.md 2>&1
cat $O/r05_dkdv_energy_decomposition.md
( timeout 300 python -m pytest tests/test_gpu_rccl_world1.py -x -q -k every_schedule ) 2>&1 | tail -3
