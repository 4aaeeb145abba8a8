"""Per-kernel sums of rocprofv3 --pmc counter_collection CSVs under a directory tree, for kernels whose name contains a pattern:
   python scripts/pmc_kernel.py <dir> <pattern> -> counter: mean per dispatch"""
import csv, glob, os, sys, collections
root, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r.get("Kernel_Name", ""):
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, n) in sorted(acc.items()):
    print(f"{k:32s} {v / max(n, 1):16.0f}   ({n} dispatches)")
