#!/bin/bash
# r04_tree_top.sh: cfg4 per-pixel + literal: head (domino), toplK (BFS top K levels layout), toprf6 (same + refetch of the top 63 slots);
# variants built with mkvariant_src.sh from csrc + profiles/r04/experiments/tree_top_probe.patch.txt (topl: no flag; toprf: -DP3D_ABL_TOP_REFETCH=63)
out=$PWD/gpurun_out/ab_top; rm -rf $out; mkdir -p $out
run() { # name lib levels mode
  P3D_LIB=$2 P3D_BFS_TOP_LEVELS=$3 python3 bench.py --workload cfg4 --stack-mode $4 --no-cpu-baseline --steps 6 --warmup 3 --frames-in-flight 1 2>$out/$1.err | tail -1 > $out/$1.json; }
V=$PWD/build/variants
for rep in 1 2; do
  for mode in per_pixel literal; do
    run head_${mode}_$rep "" "" $mode
    run topl1_${mode}_$rep $V/libp3d_topl.so 1 $mode
    run topl6_${mode}_$rep $V/libp3d_topl.so 6 $mode
    run toprf6_${mode}_$rep $V/libp3d_toprf.so 6 $mode
    run topl4_${mode}_$rep $V/libp3d_topl.so 4 $mode
  done
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab_top/*.json')):
    try:
        d=json.load(open(f)); fr=d.get('frame',{})
        print('%-24s %9.1f %8.4f | %8.4f' % (f.split('/')[-1][:-5], d['value'], d['ms_per_step'], fr.get('kernel_ms') or 0))
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-300:])
PY
