#!/bin/bash
# Profile set: kernel-trace statistics of the default benchmark, counter passes of the frame chain, of the
# IBL kernels and of the one-kernel post chain, the random-line ceiling.  Everything lands under gpurun_out/r2/.
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2
bash $R/scratch/profile_bench.sh r2/prof || exit 1
bash $R/scratch/pmc_frame.sh r2/pmc_frame || exit 1
bash $R/scratch/pmc_ibl2.sh r2/pmc_ibl || exit 1
python3 $R/scratch/pmc_summary.py $R/gpurun_out/r2/pmc_ibl irradiance > $R/gpurun_out/r2/pmc_ibl_summary.txt
python3 $R/scratch/pmc_summary.py $R/gpurun_out/r2/pmc_ibl prefilter >> $R/gpurun_out/r2/pmc_ibl_summary.txt
python3 $R/scratch/pmc_summary.py $R/gpurun_out/r2/pmc_ibl brdf_lut >> $R/gpurun_out/r2/pmc_ibl_summary.txt
cat $R/gpurun_out/r2/pmc_ibl_summary.txt
timeout -k 10 120 $R/scratch/ubench/gather_bench > $R/gpurun_out/r2/gather_bench.txt 2>&1 || exit 1
tail -20 $R/gpurun_out/r2/gather_bench.txt
