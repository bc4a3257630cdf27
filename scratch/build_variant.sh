#!/bin/bash
# build_variant.sh NAME FILE.hip "-DFLAGS": libvkr_hip.so with one translation unit rebuilt under extra flags,
# written to scratch/variants/libvkr_NAME.so (select it with VKR_HIP_LIB=...).
set -e
cd "$(dirname "$0")/../vulkanrenderer_amd/csrc"
mkdir -p ../../scratch/variants
name=$1; file=$2; flags=$3
obj=../../scratch/variants/${name}_${file%.hip}.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -Wno-unused-parameter $flags -c $file -o $obj
others=$(ls vkr_*.o | grep -v "^${file%.hip}.o$")
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/variants/libvkr_${name}.so $obj $others
echo built scratch/variants/libvkr_${name}.so
