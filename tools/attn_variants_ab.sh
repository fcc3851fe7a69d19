#!/bin/bash
# Same-box A/B of builds of the library: tools/attn_variants_ab.sh cur fin0 ...   ("cur" = the shipped library, NAME = omnibiote_amd/libomnibiote_hip_NAME.so)
for r in 1 2 3; do for l in "$@"; do if [ $l = cur ]; then unset OBTE_LIB_PATH; else export OBTE_LIB_PATH=$PWD/omnibiote_amd/libomnibiote_hip_$l.so; fi; echo "$l: $(OBTE_BENCH_B=32 python tools/attn_drop_bench.py 2>&1 | grep "^dropout 0:" )"; done; done
