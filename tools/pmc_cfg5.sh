#!/bin/bash
# SQ counters and HBM traffic of the cfg 5 frame (gather kernel, perturbed fetch): tools/pmc_cfg5.sh  -> gpurun_out/pmc_cfg5_*
set -e
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
SQA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU"
SQB="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_WAVES SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"
cd /tmp
rm -rf $O/pmc_cfg5_a $O/pmc_cfg5_b $O/pmc_cfg5_f
python3 $R/tools/cfg5_probe.py 5 > $O/pmc_cfg5.log 2>&1
rocprofv3 --pmc $SQA -d $O/pmc_cfg5_a -o a --output-format csv -- python3 $R/tools/cfg5_probe.py 3 >> $O/pmc_cfg5.log 2>&1
rocprofv3 --pmc $SQB -d $O/pmc_cfg5_b -o b --output-format csv -- python3 $R/tools/cfg5_probe.py 3 >> $O/pmc_cfg5.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_cfg5_f -o f --output-format csv -- python3 $R/tools/cfg5_probe.py 3 >> $O/pmc_cfg5.log 2>&1
