#!/bin/bash
# usage: build_variant2.sh FILE NAME "-Dflags"   -> scratch/variants/libetch_NAME.so (only etch_amd/csrc/FILE.hip is recompiled)
set -e
cd "$(dirname "$0")/../.."
file=$1; name=$2; flags=$3
mkdir -p scratch/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -I etch_amd/csrc -I include $flags -c etch_amd/csrc/$file.hip -o scratch/variants/${file}_$name.o
objs=$(ls etch_amd/lib/obj/*.o | grep -v "/$file.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/variants/libetch_$name.so $objs scratch/variants/${file}_$name.o
echo built $name
