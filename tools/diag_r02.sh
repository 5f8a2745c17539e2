#!/bin/bash
# Round-2 diagnostics on the GPU box (through gpurun, from the repo root): counter list, issue-rate
# micro-benchmark, FETCH_SIZE / WRITE_SIZE calibration on known byte counts, SQ counters of the
# config-3 / config-5 kernels.  Raw output under gpurun_out/diag/.
set -e
R=$PWD
O=$R/gpurun_out/diag
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $O/counters.txt 2>&1 || true
$R/tools/micro/issue_rate > $O/issue_rate.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/calib_f -o f -- $R/tools/micro/pmc_calib > $O/calib_bytes.txt 2> $O/calib_f.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/calib_w -o w -- $R/tools/micro/pmc_calib > /dev/null 2> $O/calib_w.log
python3 $R/tools/bench_configs.py cfg3 cfg5 > $O/cfgs_plain.jsonl 2> $O/cfgs_plain.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/tools/bench_configs.py cfg3 cfg5 > $O/cfgs_kt.jsonl 2> $O/cfgs_kt.log
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq1 -o sq1 -- python3 $R/tools/bench_configs.py cfg3 cfg5 > /dev/null 2> $O/sq1.log
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq2 -o sq2 -- python3 $R/tools/bench_configs.py cfg3 cfg5 > /dev/null 2> $O/sq2.log || true
cd $R
cat $O/issue_rate.txt
cat $O/cfgs_plain.jsonl
