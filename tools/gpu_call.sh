#!/bin/bash
# one gpurun call of the round (scratch script: edited per call)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 900 python -m pytest tests -m gpu -x -q -k "graph_capture or 256_key_form or head_dim_64 or reference_fixture" --durations=8 ) > $O/pytest_call3.log 2>&1
SH="1,8192,32,8,64,1 1,8192,32,8,64,0 1,16384,32,8,64,1 2,4096,32,8,64,1 4,2048,32,8,64,1 1,8192,32,32,64,1"
{ echo "# tools/shape_sweep.py, head dim 64: the 128-key dK/dV form (RFA_DKDV_WIDE=0: rounds 2-4) vs the round-5 plan (256-key form by the shape rules)";
  echo "## RFA_DKDV_WIDE=0"; RFA_DKDV_WIDE=0 timeout 200 python tools/shape_sweep.py $SH;
  echo "## default plan"; timeout 200 python tools/shape_sweep.py $SH;
  echo "## RFA_DKDV_WIDE=0 (again: box drift)"; RFA_DKDV_WIDE=0 timeout 200 python tools/shape_sweep.py $SH; } > $O/r05_head_dim_64_wide.md 2>&1
timeout 200 python tools/graph_step.py > $O/r05_graph_step.md 2> $O/graph_step.err
timeout 120 build/tools/mfma_power_probe 6 > $O/mfma_power_probe_6s.txt 2>&1
grep -E "passed|failed|rror" $O/pytest_call3.log | tail -8; cat $O/r05_head_dim_64_wide.md; cat $O/r05_graph_step.md; tail -3 $O/graph_step.err; cat $O/mfma_power_probe_6s.txt
