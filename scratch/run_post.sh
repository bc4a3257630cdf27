#!/bin/bash
# GPU run: parity of the one-kernel bloom chain, then its time at several band heights
set -e
mkdir -p gpurun_out/r2
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_kernel or bloom or postprocess" --timeout 120 --timeout-method thread > gpurun_out/r2/post_tests.log 2>&1 || { tail -30 gpurun_out/r2/post_tests.log; exit 1; }
tail -3 gpurun_out/r2/post_tests.log
for b in 0 16 24 32 48 72; do
  VKR_BF_BAND=$b timeout -k 10 120 python -u scratch/post_bench.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2/post_bench.log
done
