#!/bin/bash
# Developer tool (GPU box, through gpurun from the repo root): HBM-side FETCH_SIZE (KiB as reported; bytes = 2 x 1024 x that
# on gfx950) and the frame time of the slice-ring kernel on the four headline frames -- cfg 3 and the 1024^3 north star,
# brick flags on and off.   gpurun -- bash tools/traffic_probe.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for cfg in "512 cfg3 kernel=2" "512 cfg3 kernel=2,bricks=0" "1024 cfg4 kernel=2" "1024 cfg4 kernel=2,bricks=0"; do
  set -- $cfg
  rm -rf $O/tp
  rocprofv3 --pmc FETCH_SIZE -d $O/tp -o f --output-format csv -- python3 $R/tools/kbench.py --volume $1 --workload $2 --frames 6 --variants $3 > $O/tp.log 2>&1
  python3 $R/tools/pmc_summary.py $O/tp "smk_k_slab<" | grep -A1 "slab<" | tr '\n' ' '; echo
  python3 $R/tools/kbench.py --volume $1 --workload $2 --frames 30 --variants $3 2>&1 | grep "ms/frame"
done
