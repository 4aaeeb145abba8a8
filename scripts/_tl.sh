set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
PULPO_W2P_KSPLIT=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-trace > $O/ks_bench_off.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-trace > $O/ks_bench_on.json 2>/dev/null
PULPO_W2P_KSPLIT=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-trace > $O/ks_bench_off2.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-trace > $O/ks_bench_on2.json 2>/dev/null
rocprofv3 --kernel-trace -d $O/p2 -o r3t --output-format csv -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-trace > /dev/null 2>&1
T=$(find $O/p2 -name "*kernel_trace.csv" | head -1)
python scripts/timeline.py $T 10 > $O/r3_timeline_b.txt 2>&1; rm -rf $O/p2
for f in off on off2 on2; do python -c "import json,sys; d=json.loads(open('$O/ks_bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done
