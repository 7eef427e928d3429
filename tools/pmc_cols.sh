set -e
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
SQA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT"
SQB="SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA"
rm -rf $O/pmc_cols_a $O/pmc_cols_b
cd /tmp
rocprofv3 --pmc $SQA -d $O/pmc_cols_a -o a --output-format csv -- python3 $R/tools/kbench.py --volume 1024 --workload cfg4 --frames 3 --variants kernel=3,bricks=0 > $O/pmc_cols.log 2>&1
rocprofv3 --pmc $SQB -d $O/pmc_cols_b -o b --output-format csv -- python3 $R/tools/kbench.py --volume 1024 --workload cfg4 --frames 3 --variants kernel=3,bricks=0 >> $O/pmc_cols.log 2>&1
cd $R
python3 tools/pmc_summary.py $O/pmc_cols_a smk_k_cols > $O/pmc_cols_sum.txt
python3 tools/pmc_summary.py $O/pmc_cols_b smk_k_cols >> $O/pmc_cols_sum.txt
cat $O/pmc_cols_sum.txt
