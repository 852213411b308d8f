set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r01g
mkdir -p $O
cd $R
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_trace.json 2> $O/trace.err
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/pmc_fetch.err
timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/pmc_write.err
timeout -k 10 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --kernel-trace -d $O/pmc_sq --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/pmc_sq.err
timeout -k 10 240 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_sq2 --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/pmc_sq2.err
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $O/trace_tri --output-format csv -- python bench.py --workload tri100k --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_tri.json 2> $O/trace_tri.err
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $O/trace_pt --output-format csv -- python bench.py --workload cornell_pt --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_pt.json 2> $O/trace_pt.err
python bench.py --steps 200 --warmup 20 > $O/bench_plain.json 2>/dev/null
cat $O/bench_plain.json
