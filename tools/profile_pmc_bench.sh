#!/bin/bash
# PMC passes over the default bench workload (one warm-up + one step, single stream, cached plans, no tuning launches):
# FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots), then the SQ counters behind MFMA utilisation per kernel.
# Usage: bash tools/profile_pmc_bench.sh <outdir> [micro-batches per pass]   (needs <outdir>/plans.json from tools/profile_round.sh, or tunes once)
set -u
OUT=${1:-gpurun_out/pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
PP=${2:-4}   # micro-batches per pass of the profiled workload (tools/profile_round.sh writes its run's choice to <outdir>/per_pass.txt)
CMD="python3 bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_other_configs --no_roofline --no_variants --pipeline_streams 1 --micro_batches_per_pass $PP --plan_cache $OUT/plans.json"
[ -f $OUT/plans.json ] || python3 bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_other_configs --no_roofline --no_variants --plan_cache $OUT/plans.json > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o x -- $CMD > $OUT/pmc_$c.log 2>&1
  echo "$c rc=$?"
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_sq -o x -- $CMD > $OUT/pmc_sq.log 2>&1
echo "SQ rc=$?"
OBTE_PMC_PER_PASS=$PP python3 tools/pmc_traffic.py $(find $OUT/pmc_FETCH_SIZE -name "*counter_collection.csv") $(find $OUT/pmc_WRITE_SIZE -name "*counter_collection.csv") $OUT/gemm_family_traffic.json | tail -12
python3 tools/pmc_summary.py --skip 0 --match _kernel $(find $OUT/pmc_sq -name "*counter_collection.csv") > $OUT/pmc_sq_summary.txt 2>&1
grep -c "==" $OUT/pmc_sq_summary.txt
