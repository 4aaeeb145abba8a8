"""rocprofv3 --kernel-trace CSV -> where the wall time of a training step goes: per queue busy time, idle gaps, overlap, and the kernels of the
longest step sorted by their share of the main queue's critical path.
usage: python scripts/timeline.py <kernel_trace.csv> [steps in trace]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:50]) for r in rows]
ev.sort()
# take the second half of the trace (steady state)
t_lo = ev[len(ev) // 2][0]
ev = [e for e in ev if e[0] >= t_lo]
t0, t1 = ev[0][0], max(e[1] for e in ev)
span = (t1 - t0) / 1e6
queues = collections.defaultdict(list)
for s, e, q, n in ev:
    queues[q].append((s, e, n))
print(f"window {span:.1f} ms, queues: { {q: len(v) for q, v in queues.items()} }")
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
allbusy = union([(s, e) for s, e, q, n in ev])
print(f"any-kernel-running {allbusy/1e6:.1f} ms = {100*allbusy/(t1-t0):.1f} % of the window (idle GPU {span - allbusy/1e6:.1f} ms)")
for q, v in queues.items():
    b = union([(s, e) for s, e, n in v])
    print(f" queue {q}: busy {b/1e6:.1f} ms ({100*b/(t1-t0):.1f} %), {len(v)} kernels")
# per kernel name: time during which it was the ONLY kind... simpler: total duration and duration overlapped with the other queue
main = max(queues, key=lambda q: len(queues[q]))
other = [(s, e) for q, v in queues.items() if q != main for s, e, n in v]
other.sort()
def overlap(s, e):
    tot = 0
    for os_, oe in other:
        if oe <= s: continue
        if os_ >= e: break
        tot += min(e, oe) - max(s, os_)
    return tot
agg = collections.defaultdict(lambda: [0, 0, 0])
for s, e, n in queues[main]:
    a = agg[n]; a[0] += 1; a[1] += e - s; a[2] += overlap(s, e)
print(f"main queue {main}: kernel, calls, total ms, of which concurrent with the other queue ms  (window = {span:.1f} ms)")
for n, (c, d, o) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"  {n:50s} {c:5d} {d/1e6:8.2f} {o/1e6:8.2f}")
# gaps on the main queue
g = sorted(queues[main]); gaps = [g[i + 1][0] - max(x[1] for x in g[: i + 1][-3:]) for i in range(len(g) - 1)]
pos = [x for x in gaps if x > 0]
print(f"main-queue gaps: {sum(pos)/1e6:.2f} ms in {len(pos)} gaps (median {sorted(pos)[len(pos)//2]/1e3:.1f} us)")
# ---- exposed time: the parts of the window in which NO matrix-bound convolution kernel runs on either queue, by what the main queue runs then
MATRIX = ("conv3d_k3_wino3", "conv3d_k3_wino2p", "conv3d_k3_wino2_", "conv3d_k3_wgrad", "conv3d_k3_mfma", "conv3d_k3_smallk")
mat = sorted((s, e) for s, e, q, n in ev if n.startswith(MATRIX))
merged = []
for s, e in mat:
    if merged and s <= merged[-1][1]: merged[-1][1] = max(merged[-1][1], e)
    else: merged.append([s, e])
holes = [(t0, merged[0][0])] + [(merged[i][1], merged[i + 1][0]) for i in range(len(merged) - 1)] + [(merged[-1][1], t1)]
holes = [(s, e) for s, e in holes if e > s]
hole_ms = sum(e - s for s, e in holes) / 1e6
print(f"no matrix kernel on either queue: {hole_ms:.2f} ms of {span:.1f} ms ({hole_ms / nsteps * 2:.2f} ms per step if the window holds {nsteps / 2:.1f} steps), {len(holes)} holes")
import bisect
hs = [h[0] for h in holes]
def in_holes(s, e):
    tot = 0
    i = max(0, bisect.bisect_right(hs, s) - 1)
    while i < len(holes) and holes[i][0] < e:
        tot += max(0, min(e, holes[i][1]) - max(s, holes[i][0])); i += 1
    return tot
exp = collections.defaultdict(lambda: [0, 0])
covered = []
for s, e, q, n in ev:
    if n.startswith(MATRIX): continue
    d = in_holes(s, e)
    if d > 0:
        exp[n][0] += 1; exp[n][1] += d; covered.append((s, e))
print("  kernel, launches touching a hole, ms inside holes (kernels on both queues; concurrent ones count twice)")
for n, (c, d) in sorted(exp.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"  {n:50s} {c:5d} {d/1e6:8.2f}")
# idle inside holes
cov = 0
if covered:
    covered.sort(); cm = []
    for s, e in covered:
        if cm and s <= cm[-1][1]: cm[-1][1] = max(cm[-1][1], e)
        else: cm.append([s, e])
    cov = sum(in_holes(s, e) for s, e in cm)
print(f"  nothing running at all inside the holes: {hole_ms - cov/1e6:.2f} ms")
