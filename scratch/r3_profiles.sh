#!/bin/bash
# Profile set of this round: kernel-trace statistics of the default benchmark, counter passes of the frame chain and of
# the IBL kernels.  Everything lands under gpurun_out/r3/.
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3
bash $R/scratch/profile_bench.sh r3/prof || exit 1
bash $R/scratch/pmc_frame.sh r3/pmc_frame || exit 1
bash $R/scratch/pmc_ibl3.sh r3/pmc_ibl || exit 1
for k in equirect footprint_tiles irradiance_rows irradiance_regions_sum irradiance_texel prefilter brdf_lut; do
  echo "== $k"; python3 $R/scratch/pmc_summary.py $R/gpurun_out/r3/pmc_ibl $k
done > $R/gpurun_out/r3/pmc_ibl_summary.txt
for k in shade_kernel bloom_post_fused; do
  echo "== $k"; python3 $R/scratch/pmc_summary.py $R/gpurun_out/r3/pmc_frame $k
done > $R/gpurun_out/r3/pmc_frame_summary.txt
tail -40 $R/gpurun_out/r3/pmc_ibl_summary.txt
