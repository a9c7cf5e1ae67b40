#!/bin/bash
# build_variants/<name>.so = the library with ONE translation unit compiled under extra -D switches; every other object is the one of
# the regular in-tree build (run hiop_amd/build.py first).   scripts/build_variant_file.sh <name> <file.hip> [-DMACRO=1 ...]
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/.."
mkdir -p build_variants /tmp/hiopamd_variants
stem=$(basename $src .hip)
obj=/tmp/hiopamd_variants/${stem}_$name.o
extra=""
[ "$stem" = "ldlt" ] && extra="-Xclang -target-feature -Xclang -load-store-opt -mllvm -amdgpu-mfma-vgpr-form"
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -ffp-contract=on $extra "$@" -c hiop_amd/csrc/$stem.hip -o $obj
objs=$(ls hiop_amd/build/*.o | grep -v "/$stem.o")
hipcc -shared -fPIC --offload-arch=gfx950 $objs $obj -L/opt/rocm/lib -lrccl -lrocprofiler-sdk-roctx -Wl,-rpath,/opt/rocm/lib -o build_variants/$name.so
echo "built build_variants/$name.so ($stem: $*)"
