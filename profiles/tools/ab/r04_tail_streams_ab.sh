#!/bin/bash
out=$PWD/gpurun_out/ab_tail; rm -rf $out; mkdir -p $out
run() { python3 bench.py --no-cpu-baseline --steps $2 --warmup $3 --tail-streams $4 2>$out/$1.err | tail -1 > $out/$1.json; }
for rep in 1 2 3; do
  run tail_1000_$rep 1000 20 2
  run whole_1000_$rep 1000 20 0
  run tail_20_$rep 20 5 2
  run whole_20_$rep 20 5 0
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab_tail/*.json')):
    d=json.load(open(f)); print('%-16s %9.1f %8.4f %s' % (f.split('/')[-1][:-5], d['value'], d['ms_per_step'], d['config']['streams']))
PY
