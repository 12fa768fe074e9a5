#!/bin/bash
# the -DW5_TIME build of the library for scripts/w5_phase_times.py (dev tool; the product library is unaffected)
set -e
cd "$(dirname "$0")/.."
O=vanishing_points_2017_amd/csrc/_obj
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DW5_TIME -c vanishing_points_2017_amd/csrc/vpk_cnn.hip -o /tmp/vpk_cnn_w5time.o
hipcc --offload-arch=gfx950 -shared -fPIC $O/vpk_core.o $O/vpk_em.o /tmp/vpk_cnn_w5time.o $O/vpk_raster.o $O/vpk_horizon.o $O/vpk_pipeline.o $O/vpk_lsd.o -o scripts/libvpk_w5time.so
