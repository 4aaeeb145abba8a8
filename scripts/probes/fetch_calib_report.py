"""rocprofv3 --pmc FETCH_SIZE output of scripts/probes/fetch_calib -> bytes read / FETCH_SIZE per kernel (the factor scripts/pmc_traffic.py applies).
usage: python scripts/probes/fetch_calib_report.py <rocprofv3 output dir> [out.md]"""
import csv, glob, sys
from collections import defaultdict

S, C = 160, 32
_H = (S // 6) * (S // 10) * (S // 10) * 6 * 10 * 10 * C * 4.0
EXPECT = {"halo_gather_kernel<0>": _H, "halo_gather_kernel<1>": _H, "stream_read_kernel": S ** 3 * C * 4.0}
tot, cnt = defaultdict(float), defaultdict(int)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            tot[k] += float(r["Counter_Value"]) * 1024
            cnt[k] += 1
lines = ["# FETCH_SIZE calibration (scripts/probes/fetch_calib.hip, one MI355X)\n",
         "| kernel | launches | bytes read per launch (known) | FETCH_SIZE per launch | bytes / FETCH_SIZE |", "|---|---|---|---|---|"]
for k, exp in EXPECT.items():
    if cnt[k]:
        per = tot[k] / cnt[k]
        lines.append(f"| `{k}` | {cnt[k]} | {exp/2**20:.1f} MiB | {per/2**20:.1f} MiB | **{exp/per:.3f}** |")
text = "\n".join(lines) + "\n"
print(text)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text)
