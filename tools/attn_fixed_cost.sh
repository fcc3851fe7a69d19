#!/bin/bash
# Fixed cost (prologue + epilogue) of the attention kernels: time them with the tile loop cut to N tiles per workgroup.
for n in 0 1 2 4 8; do
  echo "tiles:$n  w8: $(OBTE_LIB_PATH=$PWD/omnibiote_amd/libomnibiote_hip_debug.so OBTE_ATTN_DEBUG=tiles:$n python tools/attn_bench.py --reps 20 2>&1 | grep attn)"
done
echo "full     w8: $(python tools/attn_bench.py --reps 20 2>&1 | grep attn)"
