"""Diagnostic ablation builds of single kernels: which phase of a kernel costs what.

    python scripts/ablate.py build                 (here: cross-compiles the variants into pulpo_amd/csrc/build/abl/)
    python scripts/ablate.py run [conv_bench args]  (on the GPU box: runs scripts/conv_bench.py once per variant)

A variant = one translation unit recompiled with -DPULPO_ABL=<n> (+ optional environment for the run), linked with the regular objects
of every other unit into its own shared library, which PULPO_HIP_LIB makes pulpo_amd._lib load.  Results of ablated kernels are garbage
by construction; only their timings mean anything."""
import os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pulpo_amd.build import CSRC, FLAGS, HIPCC, OBJ, build_library  # noqa: E402

ABL = os.path.join(OBJ, "abl")
# name: (unit, PULPO_ABL value, extra env at run time, conv_bench args)
VARIANTS = {
    "wino2_stamps": ("conv3d_wino", 9, {}, None),
    "wino2_stamps_stagger50": ("conv3d_wino", 9, {"PULPO_CONV_STAGGER": "50"}, None),
    "wino2_base": ("conv3d_wino", 0, {}, ["--only", "fwd"]),
    "wino2_stagger50": ("conv3d_wino", 0, {"PULPO_CONV_STAGGER": "50"}, ["--only", "fwd"]),
    "wgrad_base": ("conv3d_wgrad", 0, {}, ["--only", "wgrad"]),
    "wgrad_nodma": ("conv3d_wgrad", 11, {}, ["--only", "wgrad"]),
    "wgrad_nomfma": ("conv3d_wgrad", 12, {}, ["--only", "wgrad"]),
    "wgrad_noflush": ("conv3d_wgrad", 13, {}, ["--only", "wgrad"]),
    "w2_base": ("conv3d_wgrad_w2", 0, {}, ["--only", "wgrad", "--shapes", "0", "3", "8"]),
    "w2_nomfma": ("conv3d_wgrad_w2", 21, {}, ["--only", "wgrad", "--shapes", "0", "3", "8"]),
    "w2_nostage": ("conv3d_wgrad_w2", 22, {}, ["--only", "wgrad", "--shapes", "0", "3", "8"]),
    "w2_noloads": ("conv3d_wgrad_w2", 23, {}, ["--only", "wgrad", "--shapes", "0", "3", "8"]),
    "w2_nobarrier": ("conv3d_wgrad_w2", 24, {}, ["--only", "wgrad", "--shapes", "0", "3", "8"]),

}


def build():
    build_library()
    os.makedirs(ABL, exist_ok=True)
    done = {}
    for name, (unit, val, _, _) in VARIANTS.items():
        key = (unit, val)
        if key not in done:
            obj = os.path.join(ABL, f"{unit}_abl{val}.o")
            cmd = [HIPCC] + FLAGS + [f"-DPULPO_ABL={val}", "-c", os.path.join(CSRC, unit + ".hip"), "-o", obj]
            subprocess.run(cmd, check=True)
            lib = os.path.join(ABL, f"lib_{unit}_abl{val}.so")
            objs = [os.path.join(OBJ, f) for f in sorted(os.listdir(OBJ)) if f.endswith(".o") and f != unit + ".o"] + [obj]
            subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
            done[key] = lib
            print("built", lib)


def run(extra):
    only = [a for a in extra if a in VARIANTS]
    extra = [a for a in extra if a not in VARIANTS]
    for name, (unit, val, env, args) in VARIANTS.items():
        if only and name not in only:
            continue
        lib = os.path.join(ABL, f"lib_{unit}_abl{val}.so")
        e = dict(os.environ, PULPO_HIP_LIB=lib, **env)
        print(f"==== {name}  ({os.path.basename(lib)} {env})", flush=True)
        if args is None:          # stamp build: scripts/stamps.py on three layer shapes
            for shape in (["32", "32", "160"], ["96", "96", "80"], ["128", "128", "40"]):
                subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "stamps.py")] + shape, env=e, check=False)
            continue
        subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "conv_bench.py")] + args + extra, env=e, check=False)
        sys.stdout.flush()


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        run(sys.argv[2:])
