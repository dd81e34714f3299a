#!/bin/bash
# A second build of the library with extra -D switches on ONE source, for A/Bs inside one gpurun call:
#   tools/build_ab_lib.sh <name> <source-stem> <flags...>   ->  celerite_amd/libclr_<name>.so  (git-ignored; CLR_LIB=... for tools/gpu_ab_*.py)
set -e
name=$1; stem=$2; shift 2
mkdir -p build/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c celerite_amd/csrc/$stem.hip -o build/ab/${stem}_$name.o
objs=$(ls build/*.o | grep -v "build/$stem.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -pthread -o celerite_amd/libclr_$name.so $objs build/ab/${stem}_$name.o
echo celerite_amd/libclr_$name.so
