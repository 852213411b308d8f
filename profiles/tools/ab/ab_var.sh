#!/bin/bash
# ab_var.sh NAME... (run from the repo root): bench loop (no CPU baseline) for a reference tree under build/r02tree (mktree.sh), HEAD and the named variant libraries, alternating
out=$PWD/gpurun_out/ab_var; rm -rf $out; mkdir -p $out
for rep in ${REPS:-1 2}; do
  (cd build/r02tree && python3 bench.py --no-cpu-baseline --steps ${STEPS:-200} --warmup 20 2>$out/r02.err | tail -1 > $out/00_r02_$rep.json)
  python3 bench.py --no-cpu-baseline --steps ${STEPS:-200} --warmup 20 2>$out/head.err | tail -1 > $out/01_head_$rep.json
  for v in "$@"; do
    P3D_LIB=$PWD/build/variants/libp3d_$v.so python3 bench.py --no-cpu-baseline --steps ${STEPS:-200} --warmup 20 2>$out/$v.err | tail -1 > $out/10_${v}_$rep.json
  done
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab_var/*.json')):
    try:
        d=json.load(open(f)); fr=d.get('frame',{})
        print('%-22s %8.1f %.4f single %.4f %.4f %.4f  pp %8.1f %.4f %s' % (f.split('/')[-1], d['value'], d['ms_per_step'], fr.get('kernel_ms'), fr.get('pass1_ms'), fr.get('handoff_ms'), d['per_pixel_stack']['value'], d['per_pixel_stack'].get('kernel_ms'), d['config'].get('frames_in_flight_check','')[-3:]))
    except Exception as e: print(f, 'ERR', e)
PY
