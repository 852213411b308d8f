#!/bin/bash
# why do the tiny round-1 launches cost the loop 7 %?  skip 4 = not launched; 8 = launched, finds an empty list; w3 = list kernel without scratch
out=$PWD/gpurun_out/r04_attrib3; rm -rf $out; mkdir -p $out
V=$PWD/build/variants
run() { P3D_LIB=$2 P3D_ABL_SKIP=$3 python3 bench.py --no-cpu-baseline --steps $4 --warmup 20 2>$out/$1.err | tail -1 > $out/$1.json; }
for rep in 1 2; do
  for steps in 1000; do
    run head_${steps}_$rep "" 0 $steps
    run abl0_${steps}_$rep $V/libp3d_abl.so 0 $steps
    run abl4_${steps}_$rep $V/libp3d_abl.so 4 $steps
    run abl8_${steps}_$rep $V/libp3d_abl.so 8 $steps
    run w5_0_${steps}_$rep $V/libp3d_w5.so 0 $steps
    run w5_4_${steps}_$rep $V/libp3d_w5.so 4 $steps
  done
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04_attrib3/*.json')):
    try:
        d=json.load(open(f)); fr=d.get('frame',{})
        print('%-22s %9.1f %8.4f | %15.4f %7.4f %7.4f | %19.1f %7.4f' % (f.split('/')[-1][:-5], d['value'], d['ms_per_step'], fr.get('kernel_ms'), fr.get('pass1_ms'), fr.get('handoff_ms'), d['per_pixel_stack']['value'], d['per_pixel_stack'].get('kernel_ms')))
    except Exception as e: print(f, 'ERR', e)
PY
