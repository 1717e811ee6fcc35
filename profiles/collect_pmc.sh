#!/bin/bash
# Collect the rocprofv3 evidence under profiles/ (run on the GPU box through gpurun):
#   pass kt     --kernel-trace --stats over the exact bench.py command line the driver runs
#   pass sq/lds/fetch/write   separate --pmc passes (counters only together with --kernel-trace, as
#               MI355X_MICROARCH.md prescribes; FETCH_SIZE and WRITE_SIZE cannot share a pass)
# over the PRODUCT path (python bench.py: 5-GEMM backward with the dS spill), a few steps each.
# Output: gpurun_out/prof/<pass>/...; summaries: gpurun_out/prof/<tag>_*.txt (+ <tag>_traffic.json, which
# carries rfa_build_id() of the librfa_hip.so the counters were collected on, read from that library HERE at
# collection time — bench.py refuses to quote the file for any other build).
#   usage: bash profiles/collect_pmc.sh r03        (then copy gpurun_out/prof/r03_* into profiles/)
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof
mkdir -p $OUT
BENCH="python $R/bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline"
SHORT="python $R/bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline --no-breakdown"
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- $BENCH > $OUT/kt.log 2>&1
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o $name -- $SHORT > $OUT/$name.log 2>&1; }
run sq    SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16
run lds   SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $R
{ echo "# rocprofv3 --kernel-trace --stats -- $BENCH"; python profiles/summarize_rocpd.py $(find $OUT/kt -name "*_results.db" | head -1); grep '^{' $OUT/kt.log | tail -1; } > $OUT/${TAG}_bench_kernel_trace_stats.txt 2>&1
{ echo "# separate rocprofv3 --pmc passes over: $SHORT   (per-dispatch averages per counter instance)";
  for p in sq lds fetch write; do echo "== pass: $p"; python profiles/summarize_rocpd.py $(find $OUT/$p -name "*_results.db" | head -1) --pmc | sed -n '/per-dispatch counter averages/,$p' | tail -n +2; done; } > $OUT/${TAG}_pmc_counters.txt 2>&1
LIBSHA=$(cd $R && python -c "import sys; sys.path.insert(0, 'ring-flash-attention_amd'); from ring_flash_attn import _C; print(_C.load().rfa_build_id().decode())")
python profiles/make_traffic.py $OUT/${TAG}_pmc_counters.txt $OUT/${TAG}_traffic.json $LIBSHA
tail -5 $OUT/${TAG}_bench_kernel_trace_stats.txt; cat $OUT/${TAG}_traffic.json
