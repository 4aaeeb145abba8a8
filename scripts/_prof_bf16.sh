set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
rm -rf $O/pq
rocprofv3 --kernel-trace --stats -d $O/pq -o q --output-format csv -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-trace --precision bf16 --data oasis > /dev/null 2>&1
F=$(find $O/pq -name "*kernel_stats.csv" | head -1)
python scripts/summarize_profile.py $F $O/q_bf16_summary.md 10 "Round 3 - bf16-operand mode (python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-trace --precision bf16 --data oasis)" > /dev/null
T=$(find $O/pq -name "*kernel_trace.csv" | head -1)
python scripts/timeline.py $T 10 > $O/q_bf16_timeline.txt 2>&1
rm -rf $O/pq
