#!/bin/bash
# SQ counter passes over tools/bench_configs.py <configs...> (GPU box, through gpurun):
#   bash tools/diag_pmc.sh <tag> <config> [config...]
# Raw output under gpurun_out/pmc_<tag>/; tools/diag_pmc_report.py prints per-kernel means.
set -e
TAG=$1; shift
R=$PWD
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/p1 -o p1 -- python3 $R/tools/bench_configs.py "$@" > /dev/null 2> $O/p1.log
rocprofv3 --pmc SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_IFETCH --kernel-trace --output-format csv -d $O/p2 -o p2 -- python3 $R/tools/bench_configs.py "$@" > /dev/null 2> $O/p2.log
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/p3 -o p3 -- python3 $R/tools/bench_configs.py "$@" > /dev/null 2> $O/p3.log
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $O/p4 -o p4 -- python3 $R/tools/bench_configs.py "$@" > /dev/null 2> $O/p4.log || true
cd $R
python3 tools/diag_pmc_report.py $TAG
