#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel stats of the default bench workload (single stream, tuned plans cached so
# that no tuning launch is in the trace) + PMC passes over the attention kernels.  Usage: bash tools/profile_round.sh <outdir>
set -u
OUT=${1:-gpurun_out/prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_other_configs --no_variants --no_roofline --plan_cache $OUT/plans.json > $OUT/plan_run.log 2>&1
echo "plans cached: $(wc -c < $OUT/plans.json) bytes"
# the micro-batches per pass that run chose at start-up: the traced run takes it as given (no selection steps in the trace)
PP=$(python3 -c "import json; print(json.loads(open('$OUT/plan_run.log').read().strip().splitlines()[-1])['config']['micro_batches_per_pass'])" 2>/dev/null || echo 1)
echo "micro-batches per pass: $PP"
echo $PP > $OUT/per_pass.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o bench -- python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_other_configs --no_variants --pipeline_streams 1 --micro_batches_per_pass $PP --plan_cache $OUT/plans.json --shapes_out $OUT/shapes.txt > $OUT/bench_line.json 2> $OUT/kt.log
echo "kernel trace rc=$?"
find $OUT/kt -name "*kernel_stats.csv" | head -2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $OUT/pmc_a -o a -- python3 tools/attn_bench.py --reps 3 > $OUT/pmc_a.log 2>&1
echo "pmc a rc=$?"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_b -o b -- python3 tools/attn_bench.py --reps 3 > $OUT/pmc_b.log 2>&1
echo "pmc b rc=$?"
python3 tools/pmc_summary.py --skip 1 --match attn $(find $OUT/pmc_a -name "*counter_collection.csv") $(find $OUT/pmc_b -name "*counter_collection.csv") > $OUT/pmc_attention.txt 2>&1
tail -5 $OUT/pmc_attention.txt
