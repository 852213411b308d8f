#!/bin/bash
out=$PWD/gpurun_out/ab_tail; rm -rf $out; mkdir -p $out
run() { python3 bench.py --no-cpu-baseline --steps $2 --warmup $3 --frames-in-flight $4 --bulk-streams $5 --tail-streams $6 2>$out/$1.err | tail -1 > $out/$1.json; }
for rep in 1 2; do
  run base_f4_$rep 1000 20 4 2 0
  for f in 5 6 7 8 10 12 16; do run t_f${f}_b2_t2_$rep 1000 20 $f 2 2; done
  run base_f4_20_$rep 20 5 4 2 0
  for f in 6 8 12; do run t_f${f}_b2_t2_20_$rep 20 5 $f 2 2; done
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab_tail/*.json')):
    try:
        d=json.load(open(f)); pp=d['per_pixel_stack']
        print('%-22s %9.1f %8.4f | alone %7.4f | pp %9.1f %7.4f | %s %s' % (f.split('/')[-1][:-5], d['value'], d['ms_per_step'], d['latency_ms_single_frame'], pp['value'], pp['ms_per_step'], d['config']['frames_in_flight_check'][-3:], d['config'].get('streams')))
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-400:])
PY
