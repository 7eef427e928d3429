#!/bin/bash
# SQ counter passes of the cfg3 frame for one library build: tools/pmc_cfg3.sh <tag> [lib.so] [volume] [workload]
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
TAG=$1
[ -n "$2" ] && export SMK_LIB=$2
VOL=${3:-512}
WL=${4:-cfg3}
SQA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT"
SQB="SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA"
rm -rf $O/pmc_${TAG}_a $O/pmc_${TAG}_b
cd /tmp
rocprofv3 --pmc $SQA -d $O/pmc_${TAG}_a -o a --output-format csv -- python3 $R/tools/kbench.py --synth 0 --volume $VOL --workload $WL --frames 3 --variants kernel=2 > $O/pmc_${TAG}.log 2>&1
rocprofv3 --pmc $SQB -d $O/pmc_${TAG}_b -o b --output-format csv -- python3 $R/tools/kbench.py --synth 0 --volume $VOL --workload $WL --frames 3 --variants kernel=2 >> $O/pmc_${TAG}.log 2>&1
cd $R
python3 tools/pmc_summary.py $O/pmc_${TAG}_a smk_k_slab > $O/pmc_${TAG}.txt
python3 tools/pmc_summary.py $O/pmc_${TAG}_b smk_k_slab >> $O/pmc_${TAG}.txt
# the raw CSVs are large: keep the summary
rm -rf $O/pmc_${TAG}_a $O/pmc_${TAG}_b
