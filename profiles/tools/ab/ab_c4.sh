#!/bin/bash
# ab_c4.sh NAME...: cfg4 per-pixel + literal (frames one at a time), product library against build/variants/libp3d_NAME.so
out=$PWD/gpurun_out/ab_c4; rm -rf $out; mkdir -p $out
run() { P3D_LIB=$2 python3 bench.py --workload cfg4 --stack-mode $3 --no-cpu-baseline --steps 6 --warmup 3 --frames-in-flight 1 2>$out/$1.err | tail -1 > $out/$1.json; }
V=$PWD/build/variants
for rep in 1 2; do
  for mode in per_pixel literal; do
    run head_${mode}_$rep "" $mode
    for v in "$@"; do run ${v}_${mode}_$rep $V/libp3d_$v.so $mode; done
  done
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab_c4/*.json')):
    try:
        d=json.load(open(f)); fr=d.get('frame',{})
        print('%-24s %9.1f %8.4f | %8.4f' % (f.split('/')[-1][:-5], d['value'], d['ms_per_step'], fr.get('kernel_ms') or 0))
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-300:])
PY
