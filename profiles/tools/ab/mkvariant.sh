#!/bin/bash
# build/mkvariant.sh NAME "EXTRA FLAGS" -> build/variants/libp3d_NAME.so (experiments only, never the product library)
set -e
R=/root/repo/p3d-raytracer_amd
N=$1; shift
mkdir -p /root/repo/build/variants
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function -Wno-unused-result -I$R/../include -I$R/host -I$R/csrc "$@" -c csrc/p3d_capi.hip -o /root/repo/build/variants/$N.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC host/scene_model.o host/accel_build.o host/host_capi.o host/p3d_error.o /root/repo/build/variants/$N.o -o /root/repo/build/variants/libp3d_$N.so
echo built $N
