#!/bin/bash
mkdir -p gpurun_out/r2
for lib in "" scratch/variants/libvkr_run64.so scratch/variants/libvkr_run128.so scratch/variants/libvkr_run640.so scratch/variants/libvkr_run2048.so; do
  VKR_HIP_LIB=$lib VKR_ONLY=plain VKR_REPS=5 VKR_ROUNDS=3 timeout -k 10 200 python scratch/shade_only.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2/shade_runs.log || exit 1
done
