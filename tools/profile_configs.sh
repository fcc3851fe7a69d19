#!/bin/bash
# Bench lines + rocprofv3 kernel stats for the other BASELINE configs (4: small ctx 4096; 5: large, one GPU's share) and the
# reference-default dropout 0.1 regime.  Usage: bash tools/profile_configs.sh <outdir>
set -u
OUT=${1:-gpurun_out/prof_cfg}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
run() {  # name, extra bench args
  local name=$1; shift
  python3 bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_other_configs --no_variants --plan_cache $OUT/plans_$name.json --shapes_out $OUT/${name}_shapes.txt "$@" > $OUT/${name}_bench_line.json 2> $OUT/${name}.err
  echo "$name bench rc=$? $(python3 -c "import json,sys; d=json.loads(open('$OUT/${name}_bench_line.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])")"
  local pp=$(python3 -c "import json; print(json.loads(open('$OUT/${name}_bench_line.json').read().strip().splitlines()[-1])['config']['micro_batches_per_pass'])" 2>/dev/null || echo 1)
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$name -o $name -- python3 bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_other_configs --no_variants --no_roofline --pipeline_streams 1 --micro_batches_per_pass $pp --plan_cache $OUT/plans_$name.json "$@" > /dev/null 2> $OUT/kt_$name.log
  python3 tools/profile_family.py $(find $OUT/kt_$name -name "*kernel_stats.csv") > $OUT/${name}_families.txt
  cat $OUT/${name}_families.txt
}
run small4k --config small4k --rows_per_rank 32
run large --config large --rows_per_rank 32
run dropout --dropout 0.1
