"""rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE [SQ_BUSY_CU_CYCLES] output -> matrix-pipe busy fraction per kernel (markdown).
usage: python scripts/mfma_busy.py <rocprofv3 output dir> <out.md> "<title>"
cycles = GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs); busy = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs)."""
import csv, glob, sys
from collections import defaultdict

csv.field_size_limit(1 << 30)
d, out, title = sys.argv[1:4]
val = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        val[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[k] += 1
rows = []
for k, v in val.items():
    if v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0 and cnt[k]:
        cyc = v["GRBM_GUI_ACTIVE"] / 8 / cnt[k]
        rows.append((v["SQ_VALU_MFMA_BUSY_CYCLES"], k, cnt[k], cyc, v["SQ_VALU_MFMA_BUSY_CYCLES"] / cnt[k] / (cyc * 1024)))
rows.sort(reverse=True)
with open(out, "w") as f:
    f.write(f"# {title}\n\nCounter collection serialises the kernels: per-kernel figures over ALL launches of the step, not the overlapped step.\n\n"
            "| kernel | launches | cycles / launch | MFMA pipe busy |\n|---|---|---|---|\n")
    for _, k, n, cyc, busy in rows[:12]:
        f.write(f"| `{k}` | {n} | {cyc/1e6:.2f} M | **{100*busy:.1f} %** |\n")
print(open(out).read())
