#!/bin/bash
# mkvariant_src.sh NAME CSRC_DIR [FLAGS] -> build/variants/libp3d_NAME.so from a patched copy of csrc/ (experiments only);
# same Makefile target as mkvariant.sh, with CSRC pointing at the copy.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
N=$1; H=$(cd "$2" && pwd); shift; shift
make -s -C "$ROOT/p3d-raytracer_amd" variant NAME="$N" CSRC="$H" EXTRA="$*"
echo built $N
