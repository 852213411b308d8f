#!/bin/bash
# ab_lib.sh NAME... : bench (loop 1000 + 20 steps) for the tree's library and build/variants/libp3d_NAME.so, alternating
out=$PWD/gpurun_out/ab_lib; rm -rf $out; mkdir -p $out
run() { P3D_LIB=$2 python3 bench.py --no-cpu-baseline --steps $3 --warmup $4 2>$out/$1.err | tail -1 > $out/$1.json; }
for rep in 1 2; do
  run head_1000_$rep "" 1000 20
  for v in "$@"; do run ${v}_1000_$rep $PWD/build/variants/libp3d_$v.so 1000 20; done
  run head_20_$rep "" 20 5
  for v in "$@"; do run ${v}_20_$rep $PWD/build/variants/libp3d_$v.so 20 5; done
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab_lib/*.json')):
    try:
        d=json.load(open(f)); fr=d.get('frame',{}); pp=d['per_pixel_stack']
        print('%-16s %9.1f %8.4f | %8.4f %7.4f %7.4f | pp %9.1f %7.4f alone %7.4f %s' % (f.split('/')[-1][:-5], d['value'], d['ms_per_step'], fr.get('kernel_ms'), fr.get('pass1_ms'), fr.get('handoff_ms'), pp['value'], pp['ms_per_step'], pp['kernel_ms'], d['config']['frames_in_flight_check'][-3:]))
    except Exception as e: print(f, 'ERR', e)
PY
