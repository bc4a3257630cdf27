#!/bin/bash
# shading time per pixel and light by frame shape
mkdir -p gpurun_out/r2
VKR_ONLY=plain VKR_REPS=5 VKR_ROUNDS=2 timeout -k 10 200 python scratch/shade_only.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2/shade_shapes.log || exit 1
for cfg in "4096 64" "8192 64" "16384 64" "4096 256"; do
  set -- $cfg
  VKR_C5=1 VKR_C5_WIDTH=$1 VKR_C5_LIGHTS=$2 VKR_REPS=3 VKR_ROUNDS=2 timeout -k 10 200 python scratch/shade_only.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2/shade_shapes.log || exit 1
done
