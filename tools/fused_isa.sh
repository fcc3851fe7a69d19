#!/bin/bash
# Compile csrc/attention_bwd_fused.hip to ISA (device only) and summarise the slice loop: tools/fused_isa.sh [label-to-dump]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/omnibiote_amd/csrc" || exit 1
mkdir -p /tmp/t/f
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -S --cuda-device-only attention_bwd_fused.hip -o /tmp/t/f/fused.s 2>&1 | grep -E "error|warning: " | grep -v hip-link | head -20
grep -E "^\s+\.(vgpr_count|vgpr_spill_count|name|private_segment_fixed_size):" /tmp/t/f/fused.s | paste - - - - | head -2
python "$ROOT/tools/isa_slots.py" /tmp/t/f/fused.s fused_kernelILi128ELi1 ${1:+--dump $1}
