#!/bin/bash
# Collect the PMC evidence under profiles/ (run on the GPU box through gpurun; separate passes, counters
# only together with --kernel-trace, as MI355X_MICROARCH.md prescribes).  Output: gpurun_out/pmc/<pass>/...
# summarise with:  python profiles/summarize_rocpd.py gpurun_out/pmc/<pass>/*/*_results.db
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
B="$R/tests/native/selftest --perf-only"
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o $name -- $B > $OUT/$name.log 2>&1; }
run sq    SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16
run lds   SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
ls -R $OUT | head -40
