#!/bin/bash
# one_kernel_regs.sh LIT [CSRC] -> VGPRs / scratch / occupancy of whitted_kernel<BVH, LDS, ..., LIT, ONE_GHOSTS> compiled on its own
# (two seconds instead of the two-minute library build).  ONE_EXTRA="-DONE_GHOSTS=false -DP3D_LIST_WAVES=5 ..." adds flags.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
L=$1; C=${2:-$ROOT/p3d-raytracer_amd/csrc}
mkdir -p $ROOT/build/one
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -I$ROOT/include -I$ROOT/p3d-raytracer_amd/host -I$C -DONE_LIT=$L $ONE_EXTRA --cuda-device-only -S $ROOT/profiles/tools/ab/one_kernel.hip -o $ROOT/build/one/one_$L.s 2>&1 | grep -E "error" -A3 | head
awk '/^_ZN3p3d14whitted_kernel/{f=1} f&&/NumVgprs|ScratchSize|Occupancy/{printf "%s ", $0} f&&/Occupancy/{print ""; exit}' $ROOT/build/one/one_$L.s
