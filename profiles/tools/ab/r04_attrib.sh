#!/bin/bash
# build/r04_attrib.sh : where the literal loop's time goes: stages of the literal frame left out one by one
# (libp3d_abl.so: P3D_ABL_SKIP 1 = no check, 2 = no redo, 4 = no round 1 / persistent; libp3d_ablnr.so: pass 1 keeps no records)
out=$PWD/gpurun_out/r04_attrib; rm -rf $out; mkdir -p $out
V=$PWD/build/variants
run() {  # name lib skip steps
  P3D_LIB=$2 P3D_ABL_SKIP=$3 python3 bench.py --no-cpu-baseline --steps $4 --warmup 20 2>$out/$1.err | tail -1 > $out/$1.json
}
for rep in 1 2; do
  for steps in 1000 20; do
    run head_${steps}_$rep "" 0 $steps
    run abl0_${steps}_$rep $V/libp3d_abl.so 0 $steps
    run abl4_${steps}_$rep $V/libp3d_abl.so 4 $steps
    run abl6_${steps}_$rep $V/libp3d_abl.so 6 $steps
    run abl7_${steps}_$rep $V/libp3d_abl.so 7 $steps
    run abl5_${steps}_$rep $V/libp3d_abl.so 5 $steps
    run ablnr7_${steps}_$rep $V/libp3d_ablnr.so 7 $steps
    echo "rep $rep steps $steps done"
  done
done
python3 - <<'PY'
import json,glob
print('%-22s %9s %8s | single: %7s %7s %7s | per-pixel %9s %7s' % ('run','Mrays/s','ms/step','frame','pass1','handoff','Mrays/s','kernel'))
for f in sorted(glob.glob('gpurun_out/r04_attrib/*.json')):
    try:
        d=json.load(open(f)); fr=d.get('frame',{})
        print('%-22s %9.1f %8.4f | %15.4f %7.4f %7.4f | %19.1f %7.4f' % (f.split('/')[-1][:-5], d['value'], d['ms_per_step'], fr.get('kernel_ms'), fr.get('pass1_ms'), fr.get('handoff_ms'), d['per_pixel_stack']['value'], d['per_pixel_stack'].get('kernel_ms')))
    except Exception as e: print(f, 'ERR', e)
PY
