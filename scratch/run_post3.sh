#!/bin/bash
# parity of the one-kernel bloom chain, then its time in several builds (scratch/variants) and band heights
mkdir -p gpurun_out/r2
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_kernel or bloom or postprocess" --timeout 120 --timeout-method thread > gpurun_out/r2/post_tests.log 2>&1 || { tail -30 gpurun_out/r2/post_tests.log; exit 1; }
tail -2 gpurun_out/r2/post_tests.log
for lib in "" $VARIANTS; do
  for b in 16 24 32; do
    VKR_HIP_LIB=$lib VKR_BF_BAND=$b timeout -k 10 120 python -u scratch/post_bench.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2/post_bench3.log || exit 1
  done
done
