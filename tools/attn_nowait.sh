#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for mode in "" nowait; do
  d=gpurun_out/anw_${mode:-full}; rm -rf $d
  OBTE_LIB_PATH=$PWD/omnibiote_amd/libomnibiote_hip_debug.so OBTE_ATTN_DEBUG=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $d -o x -- python3 tools/attn_bench.py --reps 5 > /dev/null 2>&1
  echo "== ${mode:-full}"; python3 - <<PY
import csv,glob
for r in csv.DictReader(open(glob.glob("$d/**/x_kernel_stats.csv", recursive=True)[0])):
    if "attn_" in r["Name"]:
        print("  %-40s avg %8.1f us" % (r["Name"].split("attn_")[1][:38], float(r["AverageNs"])/1e3))
PY
done
