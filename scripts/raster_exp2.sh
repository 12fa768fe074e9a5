#!/bin/bash
# timing experiments: sed-patch a constant in vpk_raster.hip on the GPU box, rebuild, time (dev tool).  args: "sed-expr" ...
cd $GRAFT_REPO_ROOT
P=vanishing_points_2017_amd
cp $P/csrc/vpk_raster.hip /tmp/vpk_raster.orig
for v in "$@"; do
  cp /tmp/vpk_raster.orig $P/csrc/vpk_raster.hip
  sed -i "$v" $P/csrc/vpk_raster.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-result -Wno-unused-variable -ffp-contract=off -c $P/csrc/vpk_raster.hip -o $P/csrc/_obj/vpk_raster.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $P/csrc/_obj/*.o -o $P/libvpk.so || exit 1
  echo "== variant: $v"
  VPK_RASTER_TIMES=1 python3 scripts/time_raster.py 2>&1 | grep -v amdgpu.ids | sed -n "2p;12p;\$p" | cut -c1-135
done
cp /tmp/vpk_raster.orig $P/csrc/vpk_raster.hip
