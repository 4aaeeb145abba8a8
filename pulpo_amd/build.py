"""Build libpulpo_hip.so (gfx950 only) in-tree with hipcc.

    python -m pulpo_amd.build [--force]

hipcc cross-compiles without a GPU; the built .so sits next to the sources (pulpo_amd/csrc/libpulpo_hip.so) so it
travels to the GPU box with the tree.
"""
from __future__ import annotations

import concurrent.futures as cf
import glob
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libpulpo_hip.so")
OBJ = os.path.join(CSRC, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-Wno-inline-asm"]


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = True) -> str:
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h")))
    # (runtime.hip takes PULPO_ABI_VERSION from the public header: a version bump must rebuild it)
    hdrs.append(os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "pulpo_hip.h"))
    if not srcs:
        raise RuntimeError("no HIP sources under " + CSRC)
    os.makedirs(OBJ, exist_ok=True)
    jobs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o))
    objs = [os.path.join(OBJ, os.path.basename(s)[:-4] + ".o") for s in srcs]

    def compile_one(job):
        s, o = job
        cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return s, r.returncode, r.stdout + r.stderr

    if jobs:
        if verbose:
            print(f"[pulpo_amd.build] compiling {len(jobs)} HIP source(s) for gfx950", file=sys.stderr)
        with cf.ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for s, rc, out in ex.map(compile_one, jobs):
                if rc != 0:
                    raise RuntimeError(f"hipcc failed on {s}:\n{out}")
                if out.strip() and verbose:
                    print(out, file=sys.stderr)
    if jobs or force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
        if verbose:
            print(f"[pulpo_amd.build] linked {LIB}", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
