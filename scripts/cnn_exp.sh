#!/bin/bash
# timing experiments on the CNN: sed-patch vpk_cnn.hip on the GPU box, rebuild, time the forward (dev tool).  args: "sed-expr" ...
cd $GRAFT_REPO_ROOT
P=vanishing_points_2017_amd
cp $P/csrc/vpk_cnn.hip /tmp/vpk_cnn.orig
for v in "$@"; do
  cp /tmp/vpk_cnn.orig $P/csrc/vpk_cnn.hip
  sed -i "$v" $P/csrc/vpk_cnn.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-result -c $P/csrc/vpk_cnn.hip -o $P/csrc/_obj/vpk_cnn.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $P/csrc/_obj/*.o -o $P/libvpk.so || exit 1
  echo "== variant: $v"
  python3 scripts/time_cnn.py --passes 14 102 2>&1 | tail -2 | head -1 | cut -c1-250
done
cp /tmp/vpk_cnn.orig $P/csrc/vpk_cnn.hip
