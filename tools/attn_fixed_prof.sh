#!/bin/bash
# Kernel durations (rocprofv3 kernel trace: no launch latency in them) of the attention kernels with the tile loop cut to N tiles.
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for n in 0 4 full; do
  d=gpurun_out/afp_$n; rm -rf $d
  if [ $n = full ]; then env_s=""; else env_s="tiles:$n"; fi
  OBTE_LIB_PATH=$PWD/omnibiote_amd/libomnibiote_hip_debug.so OBTE_ATTN_DEBUG=$env_s rocprofv3 --kernel-trace --stats --output-format csv -d $d -o x -- python3 tools/attn_bench.py --reps 5 > /dev/null 2>&1
  echo "== tiles $n"; python3 - <<PY
import csv,glob
for r in csv.DictReader(open(glob.glob("$d/**/x_kernel_stats.csv", recursive=True)[0])):
    if "attn_" in r["Name"]:
        print("  %-60s calls %4s avg %8.1f us" % (r["Name"].split("(")[0][-58:], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
