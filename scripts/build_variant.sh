#!/bin/bash
# bash scripts/build_variant.sh NAME FILE.hip "-DFLAG ..."  -> build/variants/NAME.so: the library with FILE.hip compiled with the extra flags
# (diagnostic / A-B builds for LVC_AMD_LIB; the other objects come from build/csrc)
set -e
name=$1; src=$2; flags=$3
mkdir -p build/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $flags -c lvc_amd/csrc/$src -o build/variants/$name.o
objs=$(ls build/csrc/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/$name.so $objs build/variants/$name.o
echo build/variants/$name.so
