#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   (PMC and kernel-trace passes skip the secondary legs: --no-extra)
#   kernel-trace stats, FETCH_SIZE / WRITE_SIZE passes of the default bench command, SQ counter
#   passes of the two ray-march workloads, and a plain bench line.  Outputs under gpurun_out/.
#   tools/profile_round.sh flags     the product's default (empty-space skipping on) + the plain bench line
#   tools/profile_round.sh noflags   the same passes with option "bricks" 0: the streaming kernel's own counters
#   tools/profile_round.sh cols      the column-stream kernel (option kernel = 3) on the north-star frame, flags off
set -e
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
PART=${1:-flags}
SQA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT"
SQB="SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA"
if [ "$PART" = flags ]; then
rm -rf $O/prof_final $O/pmc_final_fetch $O/pmc_final_write $O/pmc_sq_a_cfg3 $O/pmc_sq_b_cfg3 $O/pmc_sq_a_ns $O/pmc_sq_b_ns
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/prof_final -o run --output-format csv -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu --no-extra > $O/prof_final.log 2>&1
rm -rf $O/prof_extra
rocprofv3 --kernel-trace --stats -d $O/prof_extra -o run --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu > $O/prof_extra.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_final_fetch -o f --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-extra > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_final_write -o w --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-extra > $O/pmc_write.log 2>&1
rocprofv3 --pmc $SQA -d $O/pmc_sq_a_cfg3 -o a --output-format csv -- python3 $R/tools/kbench.py --volume 512 --workload cfg3 --frames 3 --variants kernel=2 > $O/pmc_sq.log 2>&1
rocprofv3 --pmc $SQB -d $O/pmc_sq_b_cfg3 -o b --output-format csv -- python3 $R/tools/kbench.py --volume 512 --workload cfg3 --frames 3 --variants kernel=2 >> $O/pmc_sq.log 2>&1
rocprofv3 --pmc $SQA -d $O/pmc_sq_a_ns -o a --output-format csv -- python3 $R/tools/kbench.py --volume 1024 --workload cfg4 --frames 3 --variants kernel=2 >> $O/pmc_sq.log 2>&1
rocprofv3 --pmc $SQB -d $O/pmc_sq_b_ns -o b --output-format csv -- python3 $R/tools/kbench.py --volume 1024 --workload cfg4 --frames 3 --variants kernel=2 >> $O/pmc_sq.log 2>&1
# the dense-ramp legs' SQ counters (bench.py quotes them as roofline_valu of cfg3_dense_ramp / north_star.dense_ramp)
rm -rf $O/pmc_sq_a_cfg3_dense $O/pmc_sq_b_cfg3_dense $O/pmc_sq_a_ns_dense $O/pmc_sq_b_ns_dense
rocprofv3 --pmc $SQA -d $O/pmc_sq_a_cfg3_dense -o a --output-format csv -- python3 $R/tools/kbench.py --volume 512 --workload cfg3 --tf ramp --frames 3 --variants kernel=2 >> $O/pmc_sq.log 2>&1
rocprofv3 --pmc $SQB -d $O/pmc_sq_b_cfg3_dense -o b --output-format csv -- python3 $R/tools/kbench.py --volume 512 --workload cfg3 --tf ramp --frames 3 --variants kernel=2 >> $O/pmc_sq.log 2>&1
rocprofv3 --pmc $SQA -d $O/pmc_sq_a_ns_dense -o a --output-format csv -- python3 $R/tools/kbench.py --volume 1024 --workload cfg3 --tf ramp --frames 3 --variants kernel=2 >> $O/pmc_sq.log 2>&1
rocprofv3 --pmc $SQB -d $O/pmc_sq_b_ns_dense -o b --output-format csv -- python3 $R/tools/kbench.py --volume 1024 --workload cfg3 --tf ramp --frames 3 --variants kernel=2 >> $O/pmc_sq.log 2>&1
cd $R
python3 bench.py > $O/bench_final.json 2> $O/bench_final.err
tail -c 600 $O/bench_final.json
elif [ "$PART" = cols ]; then
# the column-stream kernel on the north-star frame, flags off: kernel trace, HBM traffic, SQ counters
cd /tmp
rm -rf $O/prof_cols $O/pmc_cols_fetch $O/pmc_cols_write $O/pmc_cols_a $O/pmc_cols_b
rocprofv3 --kernel-trace --stats -d $O/prof_cols -o run --output-format csv -- python3 $R/tools/kbench.py --volume 1024 --workload cfg4 --frames 12 --variants kernel=3,bricks=0 > $O/prof_cols.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_cols_fetch -o f --output-format csv -- python3 $R/tools/kbench.py --volume 1024 --workload cfg4 --frames 3 --variants kernel=3,bricks=0 > $O/pmc_cols_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_cols_write -o w --output-format csv -- python3 $R/tools/kbench.py --volume 1024 --workload cfg4 --frames 3 --variants kernel=3,bricks=0 > $O/pmc_cols_write.log 2>&1
rocprofv3 --pmc $SQA -d $O/pmc_cols_a -o a --output-format csv -- python3 $R/tools/kbench.py --volume 1024 --workload cfg4 --frames 3 --variants kernel=3,bricks=0 > $O/pmc_cols_sq.log 2>&1
rocprofv3 --pmc $SQB -d $O/pmc_cols_b -o b --output-format csv -- python3 $R/tools/kbench.py --volume 1024 --workload cfg4 --frames 3 --variants kernel=3,bricks=0 >> $O/pmc_cols_sq.log 2>&1
else
# the same with empty-space skipping off (option "bricks" 0): the streaming kernel's own time, traffic and counters
cd /tmp
rm -rf $O/prof_noflags $O/pmc_noflags_fetch $O/pmc_noflags_write $O/pmc_sq_a_cfg3_nf $O/pmc_sq_b_cfg3_nf $O/pmc_sq_a_ns_nf $O/pmc_sq_b_ns_nf
export SMK_BENCH_BRICKS=0
rocprofv3 --kernel-trace --stats -d $O/prof_noflags -o run --output-format csv -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu --no-extra > $O/prof_noflags.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_noflags_fetch -o f --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-extra > $O/pmc_nf_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_noflags_write -o w --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-extra > $O/pmc_nf_write.log 2>&1
unset SMK_BENCH_BRICKS
rocprofv3 --pmc $SQA -d $O/pmc_sq_a_cfg3_nf -o a --output-format csv -- python3 $R/tools/kbench.py --volume 512 --workload cfg3 --frames 3 --variants kernel=2,bricks=0 > $O/pmc_sq_nf.log 2>&1
rocprofv3 --pmc $SQB -d $O/pmc_sq_b_cfg3_nf -o b --output-format csv -- python3 $R/tools/kbench.py --volume 512 --workload cfg3 --frames 3 --variants kernel=2,bricks=0 >> $O/pmc_sq_nf.log 2>&1
rocprofv3 --pmc $SQA -d $O/pmc_sq_a_ns_nf -o a --output-format csv -- python3 $R/tools/kbench.py --volume 1024 --workload cfg4 --frames 3 --variants kernel=2,bricks=0 >> $O/pmc_sq_nf.log 2>&1
rocprofv3 --pmc $SQB -d $O/pmc_sq_b_ns_nf -o b --output-format csv -- python3 $R/tools/kbench.py --volume 1024 --workload cfg4 --frames 3 --variants kernel=2,bricks=0 >> $O/pmc_sq_nf.log 2>&1
fi
