#!/bin/bash
# Collect the rocprofv3 evidence under profiles/ (run on the GPU box through gpurun):
#   passes sq / lds / fetch / write            separate --pmc passes over the Hk = 8 headline (counters only together with
#                                              --kernel-trace, as MI355X_MICROARCH.md prescribes; FETCH_SIZE and WRITE_SIZE
#                                              cannot share a pass)
#   passes fetch_hk32 / write_hk32 / lds_hk32  the same for --kv-heads 32 (MHA)
#   pass kt / kt_hk32                          --kernel-trace --stats over the exact bench.py command line the driver runs
# over the PRODUCT path (python bench.py: 5-GEMM backward with the dS hand-off), a few steps each.
# ORDER: counters first; <tag>_traffic.json is written into profiles/ — keyed by rfa_build_id() of the librfa_hip.so the
# counters were collected on, read from that library HERE at collection time (bench.py refuses to quote the file for any
# other build) — and only THEN the kernel-trace pass runs, so that the bench line committed with the trace carries
# `roofline.traffic` of its own build instead of "stale" (round-3 review).
# Output: gpurun_out/prof/<pass>/...; summaries: gpurun_out/prof/<tag>_*.txt + <tag>_traffic.json
#   usage: bash profiles/collect_pmc.sh r06        (then copy gpurun_out/prof/r06_* into profiles/)
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof
mkdir -p $OUT
BENCH="python $R/bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline"
SHORT="python $R/bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline --no-breakdown"
run() { name=$1; extra=$2; shift 2; rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o $name -- $SHORT $extra > $OUT/$name.log 2>&1; }
run sq    "" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16
run lds   "" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE
run fetch "" FETCH_SIZE
run write "" WRITE_SIZE
run fetch_hk32 "--kv-heads 32" FETCH_SIZE
run write_hk32 "--kv-heads 32" WRITE_SIZE
run lds_hk32   "--kv-heads 32" SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
cd $R
{ echo "# separate rocprofv3 --pmc passes over: $SHORT [--kv-heads 32 for the *_hk32 passes]   (per-dispatch averages per counter instance)";
  for p in sq lds fetch write fetch_hk32 write_hk32 lds_hk32; do echo "== pass: $p"; python profiles/summarize_rocpd.py $(find $OUT/$p -name "*_results.db" | head -1) --pmc | sed -n '/per-dispatch counter averages/,$p' | tail -n +2; done; } > $OUT/${TAG}_pmc_counters.txt 2>&1
LIBSHA=$(cd $R && python -c "import sys; sys.path.insert(0, 'ring-flash-attention_amd'); from ring_flash_attn import _C; print(_C.load().rfa_build_id().decode())")
python profiles/make_traffic.py $OUT/${TAG}_pmc_counters.txt $OUT/${TAG}_traffic.json $LIBSHA
cp $OUT/${TAG}_traffic.json $R/profiles/${TAG}_traffic.json      # (on the box: the trace pass below quotes it)
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- $BENCH > $OUT/kt.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/kt_hk32 -o kt_hk32 -- $BENCH --kv-heads 32 > $OUT/kt_hk32.log 2>&1
cd $R
{ echo "# rocprofv3 --kernel-trace --stats -- $BENCH"; python profiles/summarize_rocpd.py $(find $OUT/kt -name "*_results.db" | head -1); grep '^{' $OUT/kt.log | tail -1; } > $OUT/${TAG}_bench_kernel_trace_stats.txt 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- $BENCH --kv-heads 32"; python profiles/summarize_rocpd.py $(find $OUT/kt_hk32 -name "*_results.db" | head -1); grep '^{' $OUT/kt_hk32.log | tail -1; } > $OUT/${TAG}_bench_kernel_trace_stats_hk32.txt 2>&1
tail -5 $OUT/${TAG}_bench_kernel_trace_stats.txt; python -c "import json; d=json.load(open('$OUT/${TAG}_traffic.json')); print({k: (v.get('traffic_over_algorithmic'), v.get('effective_clock_ghz_profiled'), v.get('mfma_pipe_busy')) for k, v in d.items() if isinstance(v, dict) and 'fetch_kib' in v})"
# the raw rocprofv3 databases are scratch (gpurun merges at most 64 MiB of gpurun_out/ back): keep the summaries
[ "${KEEP_TRACE:-0}" = "1" ] || rm -rf $OUT/sq $OUT/lds $OUT/fetch $OUT/write $OUT/fetch_hk32 $OUT/write_hk32 $OUT/lds_hk32 $OUT/kt $OUT/kt_hk32
