#!/bin/bash
# Timing ablations of inter_so3conv_x_kernel: rebuilds so3conv_x.o with each define set, relinks the library, times the three convs.
# usage (on the GPU box, from the repo root): bash profiles/scripts/abl_inter_x.sh "" "-DX_ABL_NODMA" ...
set -e
OBJ=etch_amd/lib/obj
for defs in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -I etch_amd/csrc -I include $defs -c etch_amd/csrc/so3conv_x.hip -o $OBJ/so3conv_x.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o etch_amd/lib/libetch_hip.so $OBJ/*.o
  echo "=== defs: [$defs]"
  python profiles/scripts/time_inter.py 10 2>&1 | grep -v amdgpu.ids | sed -e 's/max diff.*//'
done
