#!/bin/bash
# build/mkvariant_src.sh NAME CSRC_DIR [FLAGS] -> build/variants/libp3d_NAME.so from a patched copy of csrc/ (experiments only)
set -e
R=/root/repo/p3d-raytracer_amd
N=$1; H=$2; shift; shift
mkdir -p /root/repo/build/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function -Wno-unused-result -I$R/../include -I$R/host -I$H "$@" -c $H/p3d_capi.hip -o /root/repo/build/variants/$N.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $R/host/scene_model.o $R/host/accel_build.o $R/host/host_capi.o $R/host/p3d_error.o /root/repo/build/variants/$N.o -o /root/repo/build/variants/libp3d_$N.so
echo built $N
