#!/bin/bash
# one gpurun call of the round (scratch script: edited per call)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
{
for rk in 7 3; do
  echo "## rank $rk: default (library's choice)"; timeout 120 python tools/small_launch.py --rank $rk 2>/dev/null | grep "^|"
  for ns in 2 4; do
    echo "## rank $rk: RFA_FWD_FORM=8x32 RFA_FWD_KV_NSPLIT=$ns"; RFA_FWD_FORM=8x32 RFA_FWD_KV_NSPLIT=$ns timeout 120 python tools/small_launch.py --rank $rk 2>/dev/null | grep "^|"
  done
  echo "## rank $rk: RFA_FWD_FORM=8x32 (no split)"; RFA_FWD_FORM=8x32 RFA_FWD_KV_NSPLIT=1 timeout 120 python tools/small_launch.py --rank $rk 2>/dev/null | grep "^|"
done
} > $O/small_launch_forms.txt 2>&1
cat $O/small_launch_forms.txt
