# Round-4 profile recipe (run on the GPU box through gpurun):  bash profiles/r04_profile_recipe.sh <workload> <stack_mode> [steps [pmc_steps [warmup]]]
# One rocprofv3 pass per purpose — kernel trace + stats, then each PMC set on its own (FETCH_SIZE and WRITE_SIZE do not
# fit one pass; gpurun refuses --pmc together with the API trace domains) — all of the SAME command, bench.py.
# Raw output goes to gpurun_out/prof_r04/<workload>_<mode>/; profiles/summarize_r04.py condenses it into profiles/r04/.
set -e
export TMPDIR=/tmp
W=${1:-cfg2}
M=${2:-literal}
K=${3:-20}
PS=${4:-4}   # steps of the PMC passes
WU=${5:-3}   # warm-up frames of every pass (heavy workloads: 1)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r04/${W}_${M}
mkdir -p $O
cd $R
# --frames-in-flight 1: one frame at a time, so that a kernel's duration in the trace is that kernel on its own (the
# roofline block of bench.py is about the dominant kernel of ONE frame; the default timed loop overlaps two frames)
B="python3 bench.py --workload $W --stack-mode $M --no-cpu-baseline --warmup $WU --frames-in-flight 1"
timeout -k 10 ${PASS_TIMEOUT:-400} rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- $B --steps $K > $O/bench_under_rocprof.json 2> $O/trace.err
timeout -k 10 ${PASS_TIMEOUT:-400} rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch --output-format csv -- $B --steps $PS > /dev/null 2> $O/pmc_fetch.err
timeout -k 10 ${PASS_TIMEOUT:-400} rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write --output-format csv -- $B --steps $PS > /dev/null 2> $O/pmc_write.err
timeout -k 10 ${PASS_TIMEOUT:-400} rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --kernel-trace -d $O/pmc_sq --output-format csv -- $B --steps $PS > /dev/null 2> $O/pmc_sq.err
timeout -k 10 ${PASS_TIMEOUT:-400} rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_sq2 --output-format csv -- $B --steps $PS > /dev/null 2> $O/pmc_sq2.err
timeout -k 10 ${PASS_TIMEOUT:-400} rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_ACCESSES_sum --kernel-trace -d $O/pmc_tcc --output-format csv -- $B --steps $PS > /dev/null 2> $O/pmc_tcc.err || true
# condense the passes into profiles/r04/ BEFORE the closing plain run, so that its roofline block quotes these counters
# (bench.py only quotes a summary taken from exactly these kernel sources), then once more to file the plain line next to them;
# profiles/r04/ of the box travels back under gpurun_out/ (the only directory gpurun merges)
python3 profiles/summarize_r04.py $W $M > $O/summary.log 2>&1 || cat $O/summary.log
python3 bench.py --workload $W --stack-mode $M --no-cpu-baseline --warmup $WU --steps $K > $O/bench_plain.json 2> $O/bench_plain.err
python3 profiles/summarize_r04.py $W $M > $O/summary.log 2>&1 || cat $O/summary.log
python3 profiles/tools/frame_timeline.py $(ls -S $O/trace/*/*_kernel_trace.csv | head -1) > $O/frame_timeline.txt 2>&1 || true
mkdir -p $R/gpurun_out/prof_r04/summary && cp profiles/r04/${W}_${M}_* $O/frame_timeline.txt $R/gpurun_out/prof_r04/summary/ 2>/dev/null; cp $O/frame_timeline.txt $R/gpurun_out/prof_r04/summary/${W}_${M}_frame_timeline.txt 2>/dev/null
cat $O/bench_plain.json
