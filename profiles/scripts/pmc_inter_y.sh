#!/bin/bash
# SQ / TA counters of the round-5 inter conv kernel (inter_so3conv_y_kernel) and the round-4 kernels over profiles/scripts/time_inter_kq.py; separate passes.
# usage on the GPU box: bash profiles/scripts/pmc_inter_y.sh OUTDIR
set -u
O=${1:-gpurun_out/pmc_y}
mkdir -p $O
export TMPDIR=/tmp
pass() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $O/$name -o $name -- python3 profiles/scripts/time_inter_kq.py 2 > $O/$name.log 2>&1
  db=$(find $O/$name -name '*_results.db' | head -1); python3 profiles/scripts/pmc_dump.py $db inter_so3conv; rm -rf $O/$name; }
pass p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU
pass p2 SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM
pass p3 TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
pass p4 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_SALU
