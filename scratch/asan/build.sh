#!/bin/bash
# Host-side AddressSanitizer + UndefinedBehaviorSanitizer build of libvkr_hip.so and of the C++ host program (device code is
# built as usual).  usage: bash scratch/asan/build.sh   (outputs stay in scratch/asan/)
set -e
cd "$(dirname "$0")"
SRC=../../vulkanrenderer_amd/csrc
SAN="-Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer"
objs=""
for f in vkr_api vkr_shade vkr_coverage vkr_shadow vkr_gpass vkr_post vkr_ibl vkr_selftest; do
  hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden $SAN -c $SRC/$f.hip -o $f.o &
  objs="$objs $f.o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address -fsanitize=undefined -fno-gpu-sanitize -o libvkr_hip.so $objs
/opt/rocm/lib/llvm/bin/clang++ -std=c++17 -O1 -g -fsanitize=address -fsanitize=undefined -fno-omit-frame-pointer -I../../include ../../tests/cpp/renderer_roundtrip.cpp -o renderer_roundtrip \
    -L. -lvkr_hip -Wl,-rpath,'$ORIGIN' -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
rm -f *.o
ls -la
