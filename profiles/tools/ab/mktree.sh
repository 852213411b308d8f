#!/bin/bash
# build/mktree.sh COMMIT -> build/trees/COMMIT: that commit's tree with its libp3d.so built (A/B benches on one box)
set -e
c=$1; d=/root/repo/build/trees/$c
rm -rf $d; mkdir -p $d; cd /root/repo; git archive $c | tar -x -C $d
cd $d; rm -rf profiles tests/golden/fullsize tests/golden/skybox *.md *.json oracle
cd p3d-raytracer_amd && make libp3d.so > make.log 2>&1 && rm -f csrc/*.o host/*.o && echo built $c
