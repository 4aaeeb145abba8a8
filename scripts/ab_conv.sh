# same-box A/B of scripts/conv_bench.py: the committed tree in _ab/ against the working tree;  usage: bash scripts/ab_conv.sh <conv_bench options>
R=$GRAFT_REPO_ROOT
for i in 1 2; do
  echo "== A (committed)"; (cd $R/_ab && python scripts/conv_bench.py "$@" 2>/dev/null | grep -E "step-weighted|->")
  echo "== B (working tree)"; (cd $R && python scripts/conv_bench.py "$@" 2>/dev/null | grep -E "step-weighted|->")
done
