# same-box A/B of bench.py under environment settings: bash scripts/ab_env.sh <rounds> <steps> "VAR=a [VAR2=b]" "VAR=c" ...   (one line per run: setting, ms per step, roofline frac)
R=$1; S=$2; shift 2
for i in $(seq 1 $R); do
  for cfg in "$@"; do
    ( export $cfg; python bench.py --steps $S --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', round(d['ms_per_step'],3), round(d['roofline']['frac'],4))" )
  done
done
