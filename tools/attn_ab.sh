#!/bin/bash
# A/B of the attention kernels by rocprofv3 kernel durations: tools/attn_ab.sh OUTDIR "ENV_A" "ENV_B" [attn_bench args]
# e.g. tools/attn_ab.sh gpurun_out/stag "OBTE_ATTN_STAGGER=0" "OBTE_ATTN_STAGGER=1"
out=$1; ea=$2; eb=$3; shift 3
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p $out
for tag in a b; do
  if [ $tag = a ]; then e=$ea; else e=$eb; fi
  d=$out/prof_$tag; rm -rf $d
  export $e
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -o x -- python3 tools/attn_bench.py --reps 10 "$@" > $out/bench_$tag.log 2>&1
  unset ${e%%=*}
  echo "== $tag: $e"; tail -1 $out/bench_$tag.log
  python3 - <<PY
import csv, glob
f = glob.glob("$d/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if "attn" in r["Name"]:
        print(f"   {r['Name'][:90]:90s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
PY
done
