#!/bin/bash
# north_star's multi-GPU evidence in ONE command, for whoever has an 8-GPU MI355X node (the development boxes have one
# GPU; nothing here has run with N > 1 — round 5 ran it end to end at N = 1: profiles/r05_collect_scale_n1.txt):
#     bash profiles/collect_scale.sh [N=8] [TAG=r05]          (N=1: a one-GPU rehearsal on a forced one-rank RCCL group)
#   1. bench.py at N (RCCL over xGMI): the JSON line with `comm` — exchange form, the library's autotune record (ms per
#      form, max over ranks), bytes per rank, compute_only_ms / exposed_ms — `roofline` and `cpu_baseline`
#   2. tools/xgmi_probe.py: point-to-point GB/s per peer (one xGMI link each), all-gather / all-to-all / neighbour-hop
#      GB/s per rank at the schedule's message sizes, and the CONTENTION block — the 256-CU forward kernel alone, the K/V
#      all-gather alone, both at once — i.e. how much RCCL's kernels and the attention grid slow each other (SURVEY H3:
#      decides whether an SDMA / peer-copy transport is worth building)
#   3. rocprofv3 --kernel-trace --rccl-trace --hip-trace of a short bench run (trace domains only — no --pmc in this
#      pass: counters and API traces are never combined), summarised per rank: attention kernel time, RCCL kernel time,
#      and their OVERLAP on the timeline (comm hidden under compute) vs RCCL time outside any attention kernel (exposed)
# Output: gpurun_out/scale/<TAG>_*.json|txt — copy into profiles/.
set -u
N=${1:-8}
TAG=${2:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$R/gpurun_out/scale
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "$N" = "1" ]; then
  # one-GPU rehearsal of this script (profiles/r05_collect_scale_n1.txt): a ONE-rank RCCL group with the schedule
  # forced onto its multi-step path (RFA_BENCH_FORCE_RCCL: every collective, the side stream and bench.py's N > 1
  # branches run; the numbers are not measurements of anything)
  export RFA_BENCH_FORCE_RCCL=1
  CPUB="--cpu-baseline-budget-s 10"      # (the rehearsal does not need a long CPU sample)
  PROBE="python $R/tools/xgmi_probe.py"
else
  CPUB=""
  PROBE="python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29613 $R/tools/xgmi_probe.py"
fi
# bench.py launches its own N ranks (torch.distributed.run on 127.0.0.1 and a free port): the driver's command line
BENCH="python $R/bench.py --gpus $N"
# (round 6) every run first takes 2 AUDITED steps (--exchange-check-steps: each received K/V and dK/dV buffer checksummed
# against its sender, ring_flash_attn.config.exchange_check) — this is the first execution of the RCCL orderings with more
# than one rank, and a wrong exchange must fail with rank / step / buffer named instead of producing a fast wrong number
$BENCH --steps 50 --warmup 10 --exchange-check-steps 2 $CPUB > $OUT/${TAG}_bench_n${N}.json 2> $OUT/${TAG}_bench_n${N}.err
for mode in gather gather_ps ring; do
  $BENCH --steps 50 --warmup 10 --exchange-check-steps 2 --exchange $mode --no-cpu-baseline > $OUT/${TAG}_bench_n${N}_${mode}.json 2>> $OUT/${TAG}_bench_n${N}.err
done
$PROBE > $OUT/${TAG}_xgmi_probe_n${N}.json 2> $OUT/${TAG}_xgmi_probe_n${N}.err
rm -rf $OUT/trace
rocprofv3 --kernel-trace --rccl-trace --hip-trace -d $OUT/trace -o trace -- $BENCH --steps 6 --warmup 2 --no-cpu-baseline --no-breakdown > $OUT/trace.log 2>&1
cd $R
python profiles/summarize_overlap.py $OUT/trace > $OUT/${TAG}_overlap_n${N}.txt 2>&1
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/${TAG}_bench_n${N}*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        c = d.get("comm", {})
        print(f.split("/")[-1], "it/s", round(d["value"], 2), "exchange", c.get("exchange"), "exposed_ms", c.get("exposed_ms"),
              "autotune", (c.get("autotune") or {}).get("ms"), "frac", d.get("roofline", {}).get("frac"))
    except Exception as e:
        print(f, "unreadable:", e)
PY
tail -20 $OUT/${TAG}_overlap_n${N}.txt
# the raw trace (hundreds of MB with --hip-trace) is scratch: gpurun merges at most 64 MiB of gpurun_out/ back
[ "${KEEP_TRACE:-0}" = "1" ] || rm -rf $OUT/trace
