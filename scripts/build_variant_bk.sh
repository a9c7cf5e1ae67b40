#!/bin/bash
# build_variants/<name>.so = the library with csrc/ldlt_bk.hip compiled under extra -D switches (see scripts/build_variant.sh)
set -e
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p build_variants /tmp/hiopamd_variants
obj=/tmp/hiopamd_variants/ldlt_bk_$name.o
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -ffp-contract=on "$@" -c hiop_amd/csrc/ldlt_bk.hip -o $obj
objs=$(ls hiop_amd/build/*.o | grep -v "/ldlt_bk.o")
hipcc -shared -fPIC --offload-arch=gfx950 $objs $obj -L/opt/rocm/lib -lrccl -lrocprofiler-sdk-roctx -Wl,-rpath,/opt/rocm/lib -o build_variants/$name.so
echo "built build_variants/$name.so ($*)"
