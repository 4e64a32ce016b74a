#!/bin/bash
# usage (on the GPU box, from repo root): tools/prof.sh <outdir-under-gpurun_out> <bench args...>
# pass 1: kernel trace + stats; pass 2,3: PMC counters (separate runs, as the guide prescribes)
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/trace.log 2>&1 || true
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc1 -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/pmc1.log 2>&1 || true
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/pmc2.log 2>&1 || true
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/pmc3.log 2>&1 || true
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/pmc4.log 2>&1 || true
find $OUT -name "*.csv" | head -20
# texture addresser / L1 pass (k_describe, k_fem_spmv are bound there): optional as well
# (a request for more counters than one pass can hold aborts rocprofv3 and leaves the child hanging: few counters per pass, every pass under `timeout`)
timeout -k 10 300 rocprofv3 --pmc TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc6 -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/pmc6.log 2>&1 || true
timeout -k 10 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/pmc7 -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/pmc7.log 2>&1 || true
# matrix-core pass (the all-pairs matcher): optional, a missing counter must not lose the passes above
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F8 --output-format csv -d $OUT/pmc5 -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/pmc5.log 2>&1 || true
