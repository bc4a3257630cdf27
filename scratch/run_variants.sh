#!/bin/bash
# run_variants.sh SCRIPT NAME...: runs a scratch script once per library variant built by build_variant.sh
script=$1; shift
for v in "$@"; do
  echo "== $v"
  VKR_HIP_LIB=$GRAFT_REPO_ROOT/scratch/variants/libvkr_$v.so timeout -k 10 120 python3 $script 2>&1 | grep -v amdgpu.ids || exit 1
done
