#!/bin/bash
# Where the dK/dV kernel's time goes: kernel durations with parts of the tile loop removed (debug library, results wrong).
# bits: 1 no softmax arithmetic, 2 no phase-C MFMAs (dV, dK), 4 no phase-A MFMAs (S, dP), 8 no per-tile barrier
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export OBTE_LIB_PATH=$PWD/omnibiote_amd/libomnibiote_hip_debug.so
for m in 0 1 2 4 3 5 6 7 8 15; do
  d=gpurun_out/askip_$m; rm -rf $d
  OBTE_ATTN_SKIP=$m rocprofv3 --kernel-trace --stats --output-format csv -d $d -o x -- python3 tools/attn_bench.py --reps 6 > /dev/null 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("$d/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if "dkdv" in r["Name"]: print(f"skip=$m  dkdv avg {float(r['AverageNs'])/1e3:7.1f} us")
PY
done
