"""rocprofv3 --kernel-trace CSV -> the individual launch durations of the kernels whose name contains a pattern (second half of the trace)
usage: python scripts/percall.py <kernel_trace.csv> <pattern> [<pattern> ...]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]
for pat in sys.argv[2:]:
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if pat in r["Kernel_Name"]]
    grid = [(r.get("Grid_Size_X") or r.get("Grid_Size") or "?") for r in rows if pat in r["Kernel_Name"]]
    print(f"{pat}: {len(d)} launches, total {sum(d)/1e3:.3f} ms")
    by = collections.defaultdict(list)
    for g, t in zip(grid, d): by[g].append(t)
    for g, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        print(f"   grid {g:>10s}: {len(v):4d} launches, avg {sum(v)/len(v):8.1f} us, total {sum(v)/1e3:7.3f} ms")
