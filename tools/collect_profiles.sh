#!/bin/bash
# Copy the summaries of one round's profile runs from gpurun_out/ (scratch) into profiles/ (tracked):
#   tools/collect_profiles.sh r03 gpurun_out/prof_r3z gpurun_out/pmc_r3z gpurun_out/prof_cfg_r3z [final_line.json]
# (outputs of tools/profile_round.sh, tools/profile_pmc_bench.sh, tools/profile_configs.sh and a plain `python bench.py`)
set -eu
R=$1; PR=$2; PMC=$3; CFG=$4; LINE=${5:-}
cd "$(dirname "$0")/.."
hdr() { grep '^#' "$1" 2>/dev/null || true; }
cp $PR/plans.json profiles/${R}_gemm_plans_small.json
cp "$(find $PR/kt -name '*kernel_stats.csv' | head -1)" profiles/${R}_bench_small_kernel_stats.csv
cp $PR/bench_line.json profiles/${R}_bench_small_line_under_rocprof.json
cp $PR/shapes.txt profiles/${R}_bench_small_shapes.txt
cp $PR/pmc_attention.txt profiles/${R}_pmc_attention.txt
{ hdr profiles/${R}_bench_small_families.txt; python3 tools/profile_family.py profiles/${R}_bench_small_kernel_stats.csv; } > /tmp/fam.$$ && mv /tmp/fam.$$ profiles/${R}_bench_small_families.txt
cp $PMC/gemm_family_traffic.json profiles/${R}_pmc_gemm_family_traffic.json
cp $PMC/pmc_sq_summary.txt profiles/${R}_pmc_mfma_utilisation_by_kernel.txt
for n in small4k large dropout; do
  cp $CFG/${n}_bench_line.json profiles/${R}_${n}_bench_line.json
  cp $CFG/${n}_shapes.txt profiles/${R}_${n}_shapes.txt
  cp $CFG/plans_$n.json profiles/${R}_gemm_plans_$n.json
  cp "$(find $CFG/kt_$n -name '*kernel_stats.csv' | head -1)" profiles/${R}_${n}_kernel_stats.csv
  cp $CFG/${n}_families.txt profiles/${R}_${n}_families.txt
done
[ -n "$LINE" ] && cp $LINE profiles/${R}_final_bench_line.json
git status --short profiles | wc -l
