#!/bin/bash
out=$PWD/gpurun_out/ab_pt2; rm -rf $out; mkdir -p $out
for rep in 1 2; do
  for w in cfg3 cfg4; do
    st=4; [ $w = cfg4 ] && st=6
    (cd build/r02tree && python3 bench.py --workload $w --no-cpu-baseline --steps $st --warmup 2 2>$out/r02.err | tail -n 1 > $out/${w}_r02_$rep.json)
    python3 bench.py --workload $w --no-cpu-baseline --steps $st --warmup 2 2>$out/head.err | tail -n 1 > $out/${w}_head_$rep.json
  done
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab_pt2/*.json')):
    try:
        d=json.load(open(f)); print('%-24s %8.1f %.3f ms  alone %.3f  pp %s' % (f.split('/')[-1], d['value'], d['ms_per_step'], d['frame']['kernel_ms'], (d.get('per_pixel_stack') or {}).get('ms_per_step')))
    except Exception as e: print(f, 'ERR', e)
PY
