#!/bin/bash
# After `gpurun -- bash tools/profile_round.sh flags` (and `noflags`): turns what was merged into gpurun_out/ into the
# committed summaries under profiles/ (run from the repo root, here, without a GPU).   tools/collect_profiles.sh r03
set -e
TAG=${1:-r03}
O=gpurun_out
python3 tools/summarise_profiles.py $TAG $O/prof_final $O/pmc_final_fetch $O/pmc_final_write > /dev/null
cp $O/prof_extra/run_kernel_stats.csv profiles/${TAG}_kernel_stats_all_legs.csv
cp $O/bench_final.json profiles/${TAG}_bench.json
for w in cfg3 ns cfg3_dense ns_dense; do
  (echo "# rocprofv3 --pmc (two passes of 8 SQ counters) around tools/kbench.py (tools/profile_round.sh flags): per dispatch means"
   python3 tools/pmc_summary.py $O/pmc_sq_a_$w "smk_k_slab<"; python3 tools/pmc_summary.py $O/pmc_sq_b_$w "smk_k_slab<") > profiles/${TAG}_pmc_sq_$w.txt
done
if [ -d $O/prof_noflags ]; then
  python3 tools/summarise_profiles.py ${TAG}_noflags $O/prof_noflags $O/pmc_noflags_fetch $O/pmc_noflags_write > /dev/null
  for w in cfg3 ns; do
    (echo "# rocprofv3 --pmc (two passes of 8 SQ counters) around tools/kbench.py, option bricks 0 (tools/profile_round.sh noflags): per dispatch means"
     python3 tools/pmc_summary.py $O/pmc_sq_a_${w}_nf "smk_k_slab<"; python3 tools/pmc_summary.py $O/pmc_sq_b_${w}_nf "smk_k_slab<") > profiles/${TAG}_noflags_pmc_sq_$w.txt
  done
fi
