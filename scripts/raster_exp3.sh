#!/bin/bash
# timing experiments: sed-patch vpk_raster.hip AND raster_device.hpp on the GPU box, rebuild, test exactness, time (dev tool)
# args: pairs "sed-expr-for-hip" "sed-expr-for-hpp"
cd $GRAFT_REPO_ROOT
P=vanishing_points_2017_amd
cp $P/csrc/vpk_raster.hip /tmp/vpk_raster.orig; cp $P/csrc/raster_device.hpp /tmp/raster_device.orig
while [ $# -gt 1 ]; do
  cp /tmp/vpk_raster.orig $P/csrc/vpk_raster.hip; cp /tmp/raster_device.orig $P/csrc/raster_device.hpp
  sed -i "$1" $P/csrc/vpk_raster.hip; sed -i "$2" $P/csrc/raster_device.hpp
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-result -Wno-unused-variable -ffp-contract=off -c $P/csrc/vpk_raster.hip -o $P/csrc/_obj/vpk_raster.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $P/csrc/_obj/*.o -o $P/libvpk.so || exit 1
  echo "== variant: $1 | $2"
  timeout 300 python3 -m pytest tests/test_gpu_raster.py -x -q 2>&1 | tail -1
  VPK_RASTER_TIMES=1 python3 scripts/time_raster.py 2>&1 | grep -v amdgpu.ids | sed -n "2p;12p;\$p" | cut -c1-135
  shift 2
done
cp /tmp/vpk_raster.orig $P/csrc/vpk_raster.hip; cp /tmp/raster_device.orig $P/csrc/raster_device.hpp
