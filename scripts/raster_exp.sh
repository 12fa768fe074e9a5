#!/bin/bash
# timing experiments on the raster kernels: rebuild vpk_raster.o with each -D variant on the GPU box, relink, time (dev tool)
cd $GRAFT_REPO_ROOT
P=vanishing_points_2017_amd
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-result -Wno-unused-variable -ffp-contract=off $v -c $P/csrc/vpk_raster.hip -o $P/csrc/_obj/vpk_raster.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $P/csrc/_obj/*.o -o $P/libvpk.so || exit 1
  echo "== variant: $v"
  VPK_RASTER_TIMES=1 python3 scripts/time_raster.py 2 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-230
done
