#!/bin/bash
# build/one/run.sh LIT [CSRC] -> VGPRs / scratch of whitted_kernel<BVH, LDS, ..., LIT> compiled on its own
L=$1; C=${2:-/root/repo/p3d-raytracer_amd/csrc}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -I/root/repo/include -I/root/repo/p3d-raytracer_amd/host -I$C -DONE_LIT=$L $ONE_EXTRA --cuda-device-only -S /root/repo/build/one/one.hip -o /root/repo/build/one/one_$L.s 2>&1 | grep -E "error" -A3 | head
awk '/^_ZN3p3d14whitted_kernel/{f=1} f&&/NumVgprs|ScratchSize|Occupancy/{printf "%s ", $0} f&&/Occupancy/{print ""; exit}' /root/repo/build/one/one_$L.s
