#!/bin/bash
# Everything under profiles/<TAG>_* in one GPU-box command (about 11 minutes on one MI355X):
#     bash profiles/collect_round.sh [TAG=r06]
#   PMC passes + traffic file + kernel trace (collect_pmc.sh), the bench rows of the reference's tables, the compute-only
#   virtual ring, the shape sweep, the llama3 short-launch regime, the power probes, the C-ABI self test, smoke(), and the
#   whole GPU suite.  Outputs: gpurun_out/<TAG>/ — copy what is to be judged into profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import sys; sys.path.insert(0,'ring-flash-attention_amd'); from ring_flash_attn import _C; print('build id', _C.load().rfa_build_id().decode())" > $O/build_id.txt 2>&1
bash profiles/collect_pmc.sh $TAG > $O/collect_pmc.log 2>&1
cp $R/gpurun_out/prof/${TAG}_* $O/ 2>/dev/null
B="python bench.py --no-cpu-baseline"
timeout 300 python bench.py > $O/${TAG}_bench_n1_default_flags.json 2> $O/bench.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_n1_driver_command.json 2>> $O/bench.err
timeout 200 $B --kv-heads 32 > $O/${TAG}_bench_n1_mha.json 2>> $O/bench.err
timeout 200 $B --forward-only > $O/${TAG}_bench_n1_forward_only.json 2>> $O/bench.err
for wl in ring stripe ring_varlen zigzag_varlen llama3; do timeout 200 $B --workload $wl > $O/${TAG}_bench_n1_$wl.json 2>> $O/bench.err; done
{ for vw in 2 4 8; do timeout 200 $B --no-breakdown --virtual-world $vw; done; timeout 200 $B --no-breakdown --virtual-world 8 --exchange ring; } > $O/${TAG}_virtual_ring.txt 2>> $O/bench.err
timeout 300 python tools/shape_sweep.py > $O/${TAG}_shape_sweep.md 2>/dev/null
{ timeout 120 python tools/small_launch.py --rank 7 2>/dev/null | grep -v Gloo; timeout 120 python tools/small_launch.py --rank 3 2>/dev/null | grep -v Gloo; } > $O/${TAG}_small_launch_llama3.txt
timeout 200 python tools/power_probe.py --seconds 3 > $O/${TAG}_power_probe.json 2>/dev/null
mkdir -p build/tools
[ -x build/tools/mfma_power_probe ] || hipcc --offload-arch=gfx950 -O3 tools/mfma_power_probe.hip -o build/tools/mfma_power_probe -lpthread
# (round 6) the limiter read-out beside every phase incl. the MFMA probe's operand re-use patterns, the plan sweep (chosen
# vs every forced form), the host time inside Agreement.resolve() under a 28-layer stack
timeout 400 python tools/power_limiters.py --seconds 3 --json $O/${TAG}_power_limiters.json > $O/${TAG}_power_limiters_table.md 2>/dev/null
timeout 900 python tools/plan_sweep.py > $O/${TAG}_plan_sweep_final.md 2>/dev/null
# (round 6, second session) the balanced causal dK/dV schedule against the shared-range plans, the persistent forward against
# the plain forms (parity columns included)
timeout 300 python tools/bal_check.py 2>/dev/null | grep "^|" > $O/${TAG}_bal_check.md
timeout 300 python tools/fwd_persist_check.py 2>/dev/null | grep "^|" > $O/${TAG}_fwd_persist_check.md
timeout 200 python tools/agreement_stall.py > $O/${TAG}_agreement_stall.txt 2>&1
# wide head dims: the one-launch dK + dV form (head dims <= 192) against one launch per tensor (-DRFA_BG_FUSED2=0) and its variants,
# when tools/ab_variants.py built them into build/variants/
{ for v in base nofuse serial1 base nofuse serial1; do
    if [ $v = base ]; then unset RFA_LIB_PATH; elif [ -f build/variants/$v/librfa_hip.so ]; then export RFA_LIB_PATH=build/variants/$v/librfa_hip.so; else continue; fi
    echo "== $v"; timeout 200 python tools/shape_sweep.py 1,8192,20,5,192,1 1,8192,16,4,160,1 1,8192,20,5,192,0 1,8192,16,4,256,1 2>/dev/null | grep "^| [0-9]"
  done; unset RFA_LIB_PATH; } > $O/${TAG}_wide_head_dims.md 2>&1
( timeout 300 ./tests/native/selftest ) > $O/${TAG}_native_selftest.txt 2>&1
( timeout 300 python __graft_entry__.py smoke ) > $O/${TAG}_smoke.txt 2>&1
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=25 ) > $O/${TAG}_pytest_gpu.log 2>&1
tail -4 $O/${TAG}_pytest_gpu.log; cut -c1-200 $O/${TAG}_bench_n1_default_flags.json
