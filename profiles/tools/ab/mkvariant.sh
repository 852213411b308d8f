#!/bin/bash
# mkvariant.sh NAME "EXTRA FLAGS" -> build/variants/libp3d_NAME.so (experiments only, never the product library).
# Thin wrapper over the Makefile's `variant` target so that flags and architecture cannot drift from the product build.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
N=$1; shift
make -s -C "$ROOT/p3d-raytracer_amd" variant NAME="$N" EXTRA="$*"
echo built $N
