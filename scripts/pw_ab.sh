# same-box A/B of the persistent bf16 forward / data-gradient kernel (PULPO_CONV_BF16_PW = 0 off, 1 32-cout tiles, 2 all eligible shapes) on the
# bf16 configurations: config 4 step, config 5 step, config 4 inference.   usage (GPU box): bash scripts/pw_ab.sh
for m in 0 1 2; do for cfg in "4:--precision bf16 --data oasis" "5:--size 192 224 160 --levels 6 5 --precision bf16 --data oasis" "4i:--mode infer --precision bf16 --data oasis"; do
  name=${cfg%%:*}; opts=${cfg#*:}
  PULPO_CONV_BF16_PW=$m timeout -k 10 200 python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-loops --no-trace $opts 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('PW=$m cfg $name', round(d['value'],2), 'pairs/s', round(d['ms_per_step'],2), 'ms')" || exit 1
done; done
