#!/bin/bash
# one_pt_regs.sh [CSRC] -> VGPRs / scratch / occupancy of pt_kernel<BVH, LDS, no counters, ONE_SUB (4)> compiled on its own
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
C=${1:-$ROOT/p3d-raytracer_amd/csrc}
mkdir -p $ROOT/build/one
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -I$ROOT/include -I$ROOT/p3d-raytracer_amd/host -I$C $ONE_EXTRA --cuda-device-only -S $ROOT/profiles/tools/ab/one_pt_kernel.hip -o $ROOT/build/one/one_pt.s 2>&1 | grep -E "error" -A3 | head -20
awk '/^_ZN3p3d9pt_kernel/{f=1} f&&/NumVgprs|ScratchSize|Occupancy/{printf "%s ", $0} f&&/Occupancy/{print ""; exit}' $ROOT/build/one/one_pt.s
