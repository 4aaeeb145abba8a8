# same-box A/B of the register-prefetched bf16 weight-gradient staging (PULPO_WGRAD_BF16_PF=0/1) on configs 4 and 5.  usage (GPU box): bash scripts/pf_ab.sh
for m in 0 1; do for cfg in "4:--precision bf16 --data oasis" "5:--size 192 224 160 --levels 6 5 --precision bf16 --data oasis"; do
  name=${cfg%%:*}; opts=${cfg#*:}
  PULPO_WGRAD_BF16_PF=$m timeout -k 10 200 python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-loops --no-trace $opts 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('PF=$m cfg $name', round(d['value'],2), 'pairs/s', round(d['ms_per_step'],2), 'ms')" || exit 1
done; done
