#!/bin/bash
# build_variants/<name>.so = the library with csrc/ldlt.hip compiled under extra -D switches (A/B timing inside one gpurun call,
# scripts/ab_bench.sh); every other object is the one of the regular in-tree build (run hiop_amd/build.py first).
#   scripts/build_variant.sh <name> [-DMACRO=1 ...]
set -e
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p build_variants /tmp/hiopamd_variants
obj=/tmp/hiopamd_variants/ldlt_$name.o
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -ffp-contract=on -Xclang -target-feature -Xclang -load-store-opt \
      -mllvm -amdgpu-mfma-vgpr-form "$@" -c hiop_amd/csrc/ldlt.hip -o $obj
objs=$(ls hiop_amd/build/*.o | grep -v "/ldlt.o")
hipcc -shared -fPIC --offload-arch=gfx950 $objs $obj -L/opt/rocm/lib -lrccl -lrocprofiler-sdk-roctx -Wl,-rpath,/opt/rocm/lib -o build_variants/$name.so
echo "built build_variants/$name.so ($*)"
