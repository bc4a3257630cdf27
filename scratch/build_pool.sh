#!/bin/bash
# libvkr_pool.so: vkr_shadow.hip and vkr_gpass.hip rebuilt from the working tree, other objects as built.
set -e
cd "$(dirname "$0")/../vulkanrenderer_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -Wno-unused-parameter $1"
hipcc $F -c vkr_shadow.hip -o ../../scratch/variants/pool_vkr_shadow.o &
hipcc $F -c vkr_gpass.hip -o ../../scratch/variants/pool_vkr_gpass.o &
wait
others=$(ls vkr_*.o | grep -v "^vkr_gpass.o$" | grep -v "^vkr_shadow.o$")
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/variants/libvkr_pool.so ../../scratch/variants/pool_vkr_shadow.o ../../scratch/variants/pool_vkr_gpass.o $others
echo built libvkr_pool.so
