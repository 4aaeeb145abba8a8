"""rocprofv3 --kernel-trace CSV -> what every kernel family costs IN THE STREAM: its own duration plus the gap in front of it (previous kernel's
end -> this kernel's start), per step.  Gaps are split at 8 us: below = the GPU-side launch turnaround of a dependent kernel, above = the host was
not ahead (tracing slows the host down; an un-profiled fp32 step has none of these).
usage: python scripts/stream_cost.py <kernel_trace.csv> <steps in trace> [top n]"""
import csv, sys, collections

MATRIX = ("conv3d_k3", )


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:64]


rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
top = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 45
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows)
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])          # calls, duration, small gaps, large gaps
prev_end = ev[0][0]
for s, e, n in ev:
    gap = max(0, s - prev_end)
    a = agg[n]
    a[0] += 1
    a[1] += e - s
    if gap <= 8000:
        a[2] += gap
    else:
        a[3] += gap
    prev_end = max(prev_end, e)
tot = [sum(a[i] for a in agg.values()) for i in range(4)]
mat = [sum(a[i] for n, a in agg.items() if n.startswith(MATRIX)) for i in range(4)]
ms = lambda v: v / 1e6 / nsteps
print(f"{len(ev)} kernels, {nsteps:g} steps: {tot[0] / nsteps:.0f} launches per step")
print(f"per step: kernel time {ms(tot[1]):.2f} ms (matrix {ms(mat[1]):.2f}, other {ms(tot[1] - mat[1]):.2f}), launch turnaround (gaps <= 8 us) {ms(tot[2]):.2f} ms "
      f"(in front of matrix kernels {ms(mat[2]):.2f}, of others {ms(tot[2] - mat[2]):.2f}), host-bound gaps (> 8 us) {ms(tot[3]):.2f} ms")
print(f"{'kernel':64s} {'calls/step':>10s} {'ms/step':>8s} {'avg us':>7s} {'+gap ms':>8s} {'avg gap us':>10s} {'host gaps':>9s}")
for n, a in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:top]:
    print(f"{n:64s} {a[0] / nsteps:10.1f} {ms(a[1]):8.3f} {a[1] / a[0] / 1e3:7.1f} {ms(a[2]):8.3f} {a[2] / a[0] / 1e3:10.2f} {ms(a[3]):9.3f}")

# ---- the kernel sequence of the LAST complete step (between the last two adam_kernel launches): name, duration, gap in front
if "--dump" in sys.argv:
    ends = [i for i, (s, e, n) in enumerate(ev) if n.startswith("adam_kernel")]
    if len(ends) >= 2:
        lo, hi = ends[-2] + 1, ends[-1] + 1
        print(f"\n# sequence of the last step: {hi - lo} kernels")
        pe = ev[lo - 1][1]
        for s, e, n in ev[lo:hi]:
            print(f"{(e - s) / 1e3:9.1f} us  gap {max(0, s - pe) / 1e3:8.1f}  {n}  grid")
            pe = max(pe, e)
