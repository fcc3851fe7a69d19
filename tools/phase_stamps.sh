#!/bin/bash
# Per-workgroup phase stamps (s_memrealtime) of the attention backward kernels and of the K = 1024 / 4096 GEMMs, from the debug
# library (make -C omnibiote_amd/csrc debug).  Usage: bash tools/phase_stamps.sh <outfile>
OUT=${1:-gpurun_out/phase_stamps.txt}
export OBTE_LIB_PATH=$(cd "$(dirname "$0")/.." && pwd)/omnibiote_amd/libomnibiote_hip_debug.so
{
  echo "# tools/phase_stamps.sh: debug library, one launch each; times in us from the first workgroup's entry (100 MHz clock)"
  echo "# attention, B = H = 8, T = 1024, hs = 128, single-document key ranges (tools/attn_bench.py)"
  OBTE_ATTN_TIMES=1 python3 tools/attn_bench.py --reps 1 2>&1 | grep "^\[attn" | tail -4
  echo "# GEMMs of the block at M = 8192 (tools/gemm_bench.py), half-tile ring (OBTE_GEMM=v3)"
  OBTE_GEMM=v3 OBTE_GEMM_TIMES=1 python3 tools/gemm_bench.py --reps 1 --only fwd_qkv,fwd_fc,fwd_mlp,dg_mlp,dg_fc 2>&1 | grep "^\[gemm" | awk '!seen[$0]++' | awk '{k=$2" "$3" "$4" "$5; last[k]=$0} END {for (k in last) print last[k]}'
} > $OUT
cat $OUT
