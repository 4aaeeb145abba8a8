"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py -> per-kernel HBM-side bytes per launch (JSON + markdown).

usage: python scripts/pmc_traffic.py <dir of FETCH_SIZE pass> <dir of WRITE_SIZE pass> <out prefix under profiles/>

The two counters do not fit one pass on gfx950 (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2), so they come from two runs of
the same command.  Units: rocprofv3 reports both in KiB.  Correction applied as MI355X_MICROARCH.md prescribes: FETCH_SIZE reads
one half of the bytes of fully coalesced 16 B/lane streams - that applies to the LDS-DMA wgrad kernel and the elementwise
float4 kernels (factor 2 below).  The forward / data-gradient convolutions gather a 32-byte run per voxel and chunk pass, a width the
guide calls uncalibrated: scripts/probes/fetch_calib.hip reads a known byte count with exactly that pattern (profiles/r3_fetch_calibration.md)
and finds the same factor 2.000 (line fills of 128 bytes tallied at 64) - and, at the convolution's launch geometry, every line crossing
the fabric 2.9 times (the halo tiles of the 64 workgroups of an XCD exceed its 4 MiB L2 between chunk passes), which is traffic, not a
counter artefact.  Factor 2 for every kernel except the 4-byte-per-lane convolution instantiations (see FETCH_FACTOR: lower bound there).
"""
import csv, glob, json, re, sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def short(name: str) -> str:
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    if n.startswith("at::"):
        return n[:60]
    return n.split("(")[0]


def collect(d: str, counter: str):
    tot, cnt = defaultdict(float), defaultdict(int)
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            tot[k] += float(r["Counter_Value"])
            cnt[k] += 1
    return tot, cnt


# factor 2: calibrated for the 32-byte gathers of conv3d_k3_wino2p_mfma / wino2_mfma<true> (scripts/probes/fetch_calib.hip) and, by the
# guide, for 16 B/lane streams (every float4 kernel, the vectorised weight-gradient kernels).  The kernels that read 4 bytes per lane - the
# scalar direct convolution of the 2-/3-channel input layers (conv3d_k3_mfma<2|4,...>), the planar-operand instantiations (<..., false>) and
# the narrow-input weight gradient (wgrad_smallc) - were never calibrated: factor 1, i.e. their traffic figure is a LOWER bound.
FETCH_FACTOR = [(re.compile(r"conv3d_k3_mfma<(2|4),|conv3d_k3_wgrad_smallc|conv3d_k3_mfma<[^>]*false>|conv3d_k3_wino2_mfma<false|conv3d_k3_wgrad_mfma<false"), 1.0), (re.compile(r"conv3d_k3_"), 2.0)]


def main():
    fdir, wdir, out = sys.argv[1:4]
    ft, fc = collect(fdir, "FETCH_SIZE")
    wt, wc = collect(wdir, "WRITE_SIZE")
    rows = {}
    for k in ft:
        if k not in wt or fc[k] != wc[k]:
            continue
        factor = 2.0
        for pat, fac in FETCH_FACTOR:
            if pat.search(k):
                factor = fac
                break
        fetch = ft[k] * 1024 / fc[k]
        write = wt[k] * 1024 / wc[k]
        rows[k] = {"launches": fc[k], "fetch_raw_bytes": fetch, "fetch_factor": factor, "write_bytes": write,
                   "traffic_bytes": fetch * factor + write}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-trace",
               "per_launch": rows}, open(out + ".json", "w"), indent=1)
    top = sorted(rows.items(), key=lambda kv: -kv[1]["traffic_bytes"] * kv[1]["launches"])[:25]
    with open(out + ".md", "w") as f:
        f.write("# HBM-side traffic per launch (PMC), bench.py 160^3 step\n\n" + __doc__.split("\n\n", 2)[2] + "\n")
        f.write("| kernel | launches | FETCH raw MiB | factor | WRITE MiB | traffic MiB/launch |\n|---|---|---|---|---|---|\n")
        for k, r in top:
            f.write(f"| `{k}` | {r['launches']} | {r['fetch_raw_bytes']/2**20:.1f} | {r['fetch_factor']:.0f} | {r['write_bytes']/2**20:.1f} | {r['traffic_bytes']/2**20:.1f} |\n")
    print(open(out + ".md").read())


if __name__ == "__main__":
    main()
