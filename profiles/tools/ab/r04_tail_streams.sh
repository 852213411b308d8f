#!/bin/bash
# bench.py loop rate with the hand-off launches on tail streams: (frames in flight, bulk streams, tail streams[, HW queues])
out=$PWD/gpurun_out/ab_tail; rm -rf $out; mkdir -p $out
run() { # name steps warm fif bulk tail [hwq]
  if [ -n "$7" ]; then export GPU_MAX_HW_QUEUES=$7; else unset GPU_MAX_HW_QUEUES; fi
  python3 bench.py --no-cpu-baseline --steps $2 --warmup $3 --frames-in-flight $4 --bulk-streams $5 --tail-streams $6 2>$out/$1.err | tail -1 > $out/$1.json; }
for rep in 1 2; do
  run base_f4_$rep 1000 20 4 2 0
  run t_f4_b2_t2_$rep 1000 20 4 2 2
  run t_f6_b2_t2_$rep 1000 20 6 2 2
  run t_f4_b1_t3_$rep 1000 20 4 1 3
  run t_f6_b3_t1_$rep 1000 20 6 3 1
  run t_f8_b2_t6_q8_$rep 1000 20 8 2 6 8
  run t_f8_b4_t4_q8_$rep 1000 20 8 4 4 8
  run base_f4_20_$rep 20 5 4 2 0
  run t_f4_b2_t2_20_$rep 20 5 4 2 2
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab_tail/*.json')):
    try:
        d=json.load(open(f)); pp=d['per_pixel_stack']
        print('%-22s %9.1f %8.4f | alone %7.4f | pp %9.1f %7.4f | %s %s' % (f.split('/')[-1][:-5], d['value'], d['ms_per_step'], d['latency_ms_single_frame'], pp['value'], pp['ms_per_step'], d['config']['frames_in_flight_check'][-3:], d['config'].get('streams')))
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-400:])
PY
