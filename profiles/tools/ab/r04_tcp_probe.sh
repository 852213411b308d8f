#!/bin/bash
# memory-pipeline counters of the cfg4 kernel: is the vector L1 / texture addresser the busy unit?
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tcp_probe; rm -rf $O; mkdir -p $O; cd $R
rocprofv3 -L > $O/counters.txt 2>&1 || true
grep -o -E "\b(TCP|TA|TD|TCC|SQ)_[A-Z0-9_]+(_sum|_avr)?" $O/counters.txt | sort -u > $O/counter_names.txt
B="python3 bench.py --workload cfg4 --stack-mode per_pixel --no-cpu-baseline --warmup 1 --frames-in-flight 1 --steps 2"
i=0
for set in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_TOTAL_ACCESSES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum" "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace -d $O/p$i --output-format csv -- $B > /dev/null 2> $O/p$i.err || echo "set $i failed: $(tail -2 $O/p$i.err)"
done
python3 - <<'PY'
import csv,glob,collections
for d in sorted(glob.glob('gpurun_out/tcp_probe/p*/')):
    for f in glob.glob(d+'*/*counter_collection.csv'):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'whitted_kernel' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in acc.items(): print('%-44s n=%d mean=%.4g' % (k,len(v),sum(v)/len(v)))
PY
