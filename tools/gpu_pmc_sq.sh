#!/bin/bash
# GPU box: SQ counter passes (rocprofv3 --pmc, own runs with --kernel-trace only) on the Jacobian kernel of a workload.
#   tools/gpu_pmc_sq.sh <M tracks per frame> <out file>     (M = 16384: 8.4M edges, k_edge; 256: C3, k_tile)
set -u
R=$GRAFT_REPO_ROOT
M=${1:-16384}
OUT=${2:-$R/gpurun_out/pmc_sq_$M.txt}
cd /tmp && export TMPDIR=/tmp
: > $OUT
pass() {
    local name=$1; shift
    rm -rf /tmp/pmcsq_$name
    rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmcsq_$name -o runc -- python $R/tools/gpu_pmc_run.py $M 3 > /dev/null 2> /tmp/pmcsq_$name.err
    python $R/tools/pmc_summary.py $(find /tmp/pmcsq_$name -name "*results.db" | head -1) 2>&1 | grep -E "^#|k_edge|k_tile|k_stream" >> $OUT
}
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU
pass b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
pass c SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM
pass d GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F64
cat $OUT
