#!/bin/bash
# Kernel durations of the attention kernels by rocprofv3 (no launch latency in them): tools/attn_prof.sh OUTDIR [attn_bench args]
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p $out; rm -rf $out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o x -- python3 tools/attn_bench.py --reps 10 "$@" > $out/bench.log 2>&1
tail -1 $out/bench.log
python3 - <<PY
import csv, glob
f = glob.glob("$out/prof/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if "attn" in r["Name"]:
        print(f"   {r['Name'][:80]:80s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
PY
