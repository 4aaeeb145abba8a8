"""BASELINE.json configs 2-5 on one MI355X with the current build -> a markdown table (profiles/r*_configs.md).
usage: python scripts/configs_table.py <out.md>"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUNS = [
    ("2: 96^3 T4/L3 fp32 B=2", ["--size", "96", "96", "96", "--levels", "4", "3", "--batch", "2"]),
    ("3: 160^3 T5/L4 fp32 B=1 (the metric)", []),
    ("4: 160^3 T5/L4 bf16 operands + bf16 activation storage, OASIS-style pair, B=1", ["--precision", "bf16", "--data", "oasis"]),
    ("4, the step replayed from a HIP graph (--graph)", ["--precision", "bf16", "--data", "oasis", "--graph"]),
    ("4 (fp32 activation storage, the round-3 mode)", ["--precision", "bf16", "--data", "oasis", "--activations", "fp32"]),
    ("5: 192x224x160 T6/L5 bf16 operands + bf16 activation storage, training step, B=1", ["--size", "192", "224", "160", "--levels", "6", "5", "--precision", "bf16", "--data", "oasis"]),
    ("5: 192x224x160 T6/L5 bf16 operands + bf16 activation storage, 8-sample MC uncertainty maps", ["--size", "192", "224", "160", "--levels", "6", "5", "--precision", "bf16", "--data", "oasis", "--mode", "mc8"]),
    ("3, deterministic mode (--deterministic)", ["--deterministic"]),
    ("3 (inference): 160^3 predict_deterministic fp32", ["--mode", "infer"]),
    ("4 (inference): 160^3 predict_deterministic, bf16 operands + bf16 activation storage", ["--mode", "infer", "--precision", "bf16", "--data", "oasis"]),
]
rows = []
for name, extra in RUNS:
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "16", "--warmup", "3", "--no-cpu-baseline", "--no-loops"] + extra, capture_output=True, text=True, cwd=ROOT)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not line:
        rows.append(f"| {name} | failed: {r.stderr[-200:]!r} | | | | |")
        continue
    d = json.loads(line[-1])
    rf = d.get("roofline", {})
    ser = (rf.get("serialized") or {}).get("frac")
    rows.append(f"| {name} | {d['value']:.2f} | {d['ms_per_step']:.2f} | {rf.get('kernel', '')} | {rf.get('frac', float('nan')):.3f} | {'' if ser is None else f'{ser:.3f}'} | {d.get('hbm_peak_allocated_GB', float('nan')):.1f} |")
    print(rows[-1], flush=True)
with open(sys.argv[1], "w") as f:
    f.write("# BASELINE configs on one MI355X (python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-loops + the options of the config)\n\n"
            "`frac` = FLOP the dominant kernel issues on the matrix pipe over its live-bracketed time in the overlapped step, against the dense peak of its "
            "operand type (157.3 TFLOP/s fp32, 2500 bf16); `serialized` = the same bracket with the weight-gradient stream off.\n\n"
            "| config | pairs/s | ms/step | dominant kernel | roofline.frac | serialized.frac | HBM peak GB |\n|---|---|---|---|---|---|---|\n" + "\n".join(rows) + "\n")
