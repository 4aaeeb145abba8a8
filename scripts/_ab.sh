# same-box A/B: the committed tree in _ab/ against the working tree
set -e
R=$GRAFT_REPO_ROOT
N=${1:-3}
for i in $(seq $N); do
  (cd $R/_ab && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-trace 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('A(head)', round(d['ms_per_step'],3))")
  (cd $R && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-trace 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B(work)', round(d['ms_per_step'],3))")
done
