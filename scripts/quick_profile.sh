set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
rm -rf $O/pq
rocprofv3 --kernel-trace --stats -d $O/pq -o q --output-format csv -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-trace > /dev/null 2>&1
F=$(find $O/pq -name "*kernel_stats.csv" | head -1)
python scripts/summarize_profile.py $F $O/q_summary.md 10 "work tree" > /dev/null
T=$(find $O/pq -name "*kernel_trace.csv" | head -1)
python scripts/timeline.py $T 10 > $O/q_timeline.txt 2>&1
python scripts/percall.py $T ${PERCALL:-vecint_bwd_tile} > $O/q_percall.txt 2>&1
rm -rf $O/pq
head -60 $O/q_summary.md
