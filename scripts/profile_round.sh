set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
# 1. headline bench (with cpu baseline) - the line the driver would produce
python bench.py > $O/r5_bench_final.json 2> $O/r5_bench_final.err
# 2. kernel trace + stats of the same command r2 used
rocprofv3 --kernel-trace --stats -d $O/p1 -o r5 --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-loops > $O/r5_under_rocprof.json 2>/dev/null
F=$(find $O/p1 -name "*kernel_stats.csv" | head -1)
python scripts/summarize_profile.py $F $O/r5_bench160_summary.md 12 "Round 5 - rocprofv3 --kernel-trace --stats of python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-loops (2 warm-up + 5 timed + 2 untimed steps with every memory-bound launch bracketed + 3 untimed serialized-trace steps)" > /dev/null
cp $F $O/r5_bench160_kernel_stats.csv; rm -rf $O/p1
# 3. timeline of the overlapped step (no event brackets, no serialized steps)
rocprofv3 --kernel-trace -d $O/p2 -o r5t --output-format csv -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-trace --no-loops > /dev/null 2>&1
T=$(find $O/p2 -name "*kernel_trace.csv" | head -1)
python scripts/timeline.py $T 10 > $O/r5_bench160_timeline.txt 2>&1
python scripts/stream_cost.py $T 10 90 --dump > $O/r5_bench160_stream_cost.txt 2>&1; rm -rf $O/p2
# 4. PMC traffic passes
rocprofv3 --pmc FETCH_SIZE -d $O/pf --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-trace --no-loops > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pw --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-trace --no-loops > /dev/null 2>&1
python scripts/pmc_traffic.py $O/pf $O/pw $O/r5_bench160_pmc_traffic > /dev/null; rm -rf $O/pf $O/pw
# 5. matrix-pipe busy
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pm --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-trace --no-loops > /dev/null 2>&1
python scripts/mfma_busy.py $O/pm $O/r5_conv_pmc_step.md "Round 5 - matrix-pipe busy per kernel over a whole 160^3 step (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-trace --no-loops)" > /dev/null; rm -rf $O/pm
tail -c 600 $O/r5_bench_final.json
