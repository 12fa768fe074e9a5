#!/bin/bash
# Build libvpk.so on the GPU box with extra compiler flags for vpk_cnn.hip (e.g. -DSG_SMALL_SEPARATE), run a command, restore
# the default build (dev tool):  bash scripts/cnn_variant.sh "-DFLAG ..." 'command'
cd $GRAFT_REPO_ROOT
P=vanishing_points_2017_amd
cp $P/libvpk.so /tmp/libvpk.default.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-result $1 -c $P/csrc/vpk_cnn.hip -o /tmp/vpk_cnn_variant.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $P/csrc/_obj/*.o | grep -v vpk_cnn.o) /tmp/vpk_cnn_variant.o -o $P/libvpk.so || exit 1
echo "== variant: $1"
bash -c "$2"
cp /tmp/libvpk.default.so $P/libvpk.so
