"""rocprofv3 --kernel-trace --stats CSV -> markdown summary under profiles/.
usage: python scripts/summarize_profile.py <kernel_stats.csv> <out.md> <steps in trace> "<title>" """
import csv, sys
src, out, nsteps, title = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
rows = list(csv.DictReader(open(src)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(out, "w") as f:
    f.write(f"# {title}\n\n160^3 fp32, T5/L4, B=1, one MI355X; {nsteps} steps in the trace.  Source CSV next to this file.\n")
    f.write(f"\nSum of kernel durations {tot/1e6:.1f} ms = {tot/1e6/nsteps:.2f} ms/step (weight-gradient kernels run on a second stream concurrently "
            f"with the main stream, so this sum exceeds the wall time per step and overlapped kernels show stretched durations)\n\n| kernel | calls/step | ms/step | avg us | % |\n|---|---|---|---|---|\n")
    for r in rows[:45]:
        n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        n = n.split("(")[0] if not n.startswith("at::") else n[:70]
        f.write(f"| `{n}` | {int(r['Calls'])/nsteps:.1f} | {float(r['TotalDurationNs'])/1e6/nsteps:.3f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.1f} |\n")
print(open(out).read()[:1800])
