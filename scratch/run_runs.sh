#!/bin/bash
mkdir -p gpurun_out/r2
for h in 270 2160; do
for l in 1 8 16 32 64; do
  VKR_HEIGHT=$h VKR_LIGHTS=$l VKR_ONLY=plain VKR_REPS=20 VKR_ROUNDS=2 timeout -k 10 200 python scratch/shade_only.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2/shade_lights_small.log || exit 1
done; done
