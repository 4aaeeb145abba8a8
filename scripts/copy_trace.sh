set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
rm -rf $O/pmc_copy
rocprofv3 --hip-runtime-trace -d $O/pmc_copy -o mc --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-trace > /dev/null 2>&1
ls $O/pmc_copy/*
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_copy/**/*memory_copy_trace.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    print(len(rows), "copies; columns:", list(rows[0].keys()))
    c = collections.Counter((r.get("Direction"), r.get("Bytes", r.get("Size"))) for r in rows)
    for k, v in c.most_common(25): print(v, k)
f = glob.glob("gpurun_out/pmc_copy/**/*hip_api_trace.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    print(list(rows[0].keys()))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[len(rows) * 2 // 3:]
    idx = [i for i, r in enumerate(rows) if r["Function"] in ("hipMemcpyWithStream", "hipMemcpyAsync")]
    print(len(idx), "copies in the last third")
    # context: the API calls just before each copy (names only), tallied
    ctx = collections.Counter()
    for i in idx:
        prev = [rows[j]["Function"] for j in range(max(0, i - 6), i) if rows[j]["Function"] not in ("hipGetDevice", "hipSetDevice", "hipGetLastError", "hipGetDeviceCount", "hipStreamGetCaptureInfo", "hipStreamIsCapturing", "hipDevicePrimaryCtxGetState")]
        ctx[(rows[i]["Function"], tuple(prev[-3:]))] += 1
    for k, v in ctx.most_common(20): print(v, k)
    # durations
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if r["Function"] == "hipMemcpyWithStream"]
    if d: print("hipMemcpyWithStream host time: avg %.1f us, total %.2f ms" % (sum(d) / len(d) / 1e3, sum(d) / 1e6))
PY
rm -rf $O/pmc_copy
