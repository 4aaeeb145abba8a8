# same-box A/B: a copy of the committed tree in _ab/ (git archive HEAD | tar -x -C _ab; python -m pulpo_amd.build there) against the working tree;
# usage (on the GPU box): bash scripts/ab_bench.sh [repetitions]
set -e
R=$GRAFT_REPO_ROOT
N=${1:-3}
for i in $(seq $N); do
  (cd $R/_ab && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-trace 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('A(head)', round(d['ms_per_step'],3))")
  (cd $R && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-trace 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B(work)', round(d['ms_per_step'],3))")
done
