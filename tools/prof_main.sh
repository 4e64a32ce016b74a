#!/bin/bash
# usage (GPU box, repo root): tools/prof_main.sh <outdir-under-gpurun_out>
# The frames/s path alone (bench.py --no-fem: extract + match, one context on one stream, so no kernel overlaps another):
# kernel trace + the PMC passes, few counters per pass, every pass under `timeout` (a pass that asks for more counters
# than the hardware holds aborts rocprofv3 and leaves the child hanging).  tools/prof_pick.py summarises the 64-frame launches.
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
A="--steps 10 --pipeline 1 --no-fem --no-cpu-baseline --no-host-io --no-verify"
run() { name=$1; shift; timeout -k 10 240 rocprofv3 "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/bench.py $A > $OUT/$name.log 2>&1; echo "$name rc $?"; }
run trace --kernel-trace --stats
run pmc1 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
run pmc2 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE
run pmc3 --pmc FETCH_SIZE
run pmc4 --pmc WRITE_SIZE
run pmc5 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
run pmc6 --pmc TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE
run pmc7 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
