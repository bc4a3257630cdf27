#!/bin/bash
# isa_stats.sh FILE.hip KERNEL_SUBSTRING "-DFLAGS": registers, spills, code size and instruction mix of one kernel
cd "$(dirname "$0")/../vulkanrenderer_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -S --offload-device-only $3 $1 -o /tmp/isa_$$.s 2>/dev/null
awk -v k="$2" '$0 ~ "^_Z.*"k".*:"{f=1} f{print} /s_endpgm/{if(f){exit}}' /tmp/isa_$$.s > /tmp/isa_$$_k.s
grep -A12 "\.name:.*$2" /tmp/isa_$$.s | grep -E "vgpr_count|vgpr_spill|sgpr_spill|private_segment" | tr -s ' ' | tr '\n' ' '
echo " instr=$(grep -cE '^\s+[vsdg][a-z_]*_' /tmp/isa_$$_k.s) valu=$(grep -cE '^\s+v_' /tmp/isa_$$_k.s) trans=$(grep -cE 'v_(log|exp|rcp|rsq|sqrt)_f32' /tmp/isa_$$_k.s) lds=$(grep -cE '^\s+ds_' /tmp/isa_$$_k.s) vmem=$(grep -cE '^\s+(global|buffer|scratch)_' /tmp/isa_$$_k.s)"
rm -f /tmp/isa_$$.s /tmp/isa_$$_k.s
