#!/bin/bash
# mktree.sh COMMIT -> build/trees/COMMIT: that commit's tree with its libp3d.so built (A/B benches on one box)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
c=$1; d=$ROOT/build/trees/$c
rm -rf $d; mkdir -p $d; cd $ROOT; git archive $c | tar -x -C $d
cd $d; rm -rf profiles tests/golden/fullsize tests/golden/skybox *.md *.json oracle
cd p3d-raytracer_amd && make libp3d.so > make.log 2>&1 && rm -f csrc/*.o host/*.o && echo built $c
