#!/bin/bash
# shading time at the 16384 x 2048 / 256-light strip by lights per launch
mkdir -p gpurun_out/r2
for c in 0 128 64 32 16; do
  VKR_SHADE_CHUNK=$c VKR_C5=1 VKR_REPS=3 VKR_ROUNDS=2 timeout -k 10 200 python scratch/shade_only.py 2>&1 | grep -v amdgpu.ids | sed "s/^/chunk $c: /" | tee -a gpurun_out/r2/shade_chunks.log || exit 1
done
