# rocprofv3 --kernel-trace --stats of a short bench.py run -> gpurun_out/<prefix>_summary.md (kernel table), <prefix>_timeline.txt (scripts/timeline.py,
# incl. what runs while no matrix kernel is active) and <prefix>_percall.txt (scripts/percall.py for the kernels named in $PERCALL).
# usage (on the GPU box): [PERCALL="name ..."] bash scripts/quick_profile.sh [prefix=q] [bench.py options, e.g. --precision bf16 --data oasis]
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
P=${1:-q}
if [ $# -gt 0 ]; then shift; fi
rm -rf $O/pq
rocprofv3 --kernel-trace --stats -d $O/pq -o q --output-format csv -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-trace "$@" > /dev/null 2>&1
F=$(find $O/pq -name "*kernel_stats.csv" | head -1)
python scripts/summarize_profile.py $F $O/${P}_summary.md 10 "python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-trace $*" > /dev/null
T=$(find $O/pq -name "*kernel_trace.csv" | head -1)
python scripts/timeline.py $T 10 > $O/${P}_timeline.txt 2>&1
python scripts/percall.py $T ${PERCALL:-vecint_bwd_tile} > $O/${P}_percall.txt 2>&1
rm -rf $O/pq
head -40 $O/${P}_summary.md
