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
    "wino2_stamps": ("conv3d_wino", 9, {"PULPO_W2_PIPE": "0"}, None),
    "wino2_base": ("conv3d_wino", 0, {"PULPO_W2_PIPE": "0"}, ["--only", "fwd"]),
    "wgrad_base": ("conv3d_wgrad", 0, {}, ["--only", "wgrad"]),
    "wgrad_nodma": ("conv3d_wgrad", 11, {}, ["--only", "wgrad"]),
    "wgrad_nomfma": ("conv3d_wgrad", 12, {}, ["--only", "wgrad"]),
    "wgrad_noflush": ("conv3d_wgrad", 13, {}, ["--only", "wgrad"]),
    "w2_base": ("conv3d_wgrad_w2", 0, {}, ["--only", "wgrad", "--shapes", "0", "3", "8"]),
    "w2_nomfma": ("conv3d_wgrad_w2", 21, {}, ["--only", "wgrad", "--shapes", "0", "3", "8"]),
    "w2_nostage": ("conv3d_wgrad_w2", 22, {}, ["--only", "wgrad", "--shapes", "0", "3", "8"]),
    "w2_noloads": ("conv3d_wgrad_w2", 23, {}, ["--only", "wgrad", "--shapes", "0", "3", "8"]),
    "w2_nobarrier": ("conv3d_wgrad_w2", 24, {}, ["--only", "wgrad", "--shapes", "0", "3", "8"]),
    # pipelined (y, x) forward / data-gradient kernel (conv3d_wino2p.hip): PULPO_ABL is a bit mask there
    "p_base": ("conv3d_wino2p", 0, {}, ["--only", "fwd", "--shapes", "0", "3", "8"]),
    "p_noepi": ("conv3d_wino2p", 1, {}, ["--only", "fwd", "--shapes", "0", "3", "8"]),
    "p_nostage": ("conv3d_wino2p", 2, {}, ["--only", "fwd", "--shapes", "0", "3", "8"]),
    "p_noweights": ("conv3d_wino2p", 4, {}, ["--only", "fwd", "--shapes", "0", "3", "8"]),
    "p_nobarrier": ("conv3d_wino2p", 8, {}, ["--only", "fwd", "--shapes", "0", "3", "8"]),
    "p_nomfma": ("conv3d_wino2p", 16, {}, ["--only", "fwd", "--shapes", "0", "3", "8"]),
    "p_bare": ("conv3d_wino2p", 1 + 2 + 4 + 8, {}, ["--only", "fwd", "--shapes", "0", "3", "8"]),
    "p_bare_noreads": ("conv3d_wino2p", 1 + 2 + 4 + 8 + 32, {}, ["--only", "fwd", "--shapes", "0", "3", "8"]),
    "p_nocomb": ("conv3d_wino2p", 256, {}, ["--only", "fwd", "--shapes", "0", "3", "8"]),
    "p_nocomb_nostage": ("conv3d_wino2p", 256 + 2, {}, ["--only", "fwd", "--shapes", "0", "3", "8"]),
    "p_stage_loads_only": ("conv3d_wino2p", 512, {}, ["--only", "fwd", "--shapes", "0", "3", "8"]),
    "p_stage_writes_only": ("conv3d_wino2p", 1024, {}, ["--only", "fwd", "--shapes", "0", "3", "8"]),
    "p_stamps": ("conv3d_wino2p", 64, {}, None),
    "p_stamps_nostore": ("conv3d_wino2p", 64 + 128, {}, None),
    "p_nostore": ("conv3d_wino2p", 128, {}, ["--only", "fwd", "--shapes", "0", "3", "8"]),
    # F(2x2x2,3x3x3) kernel (conv3d_wino3.hip): bit mask
    "t_base": ("conv3d_wino3", 0, {}, ["--only", "fwd", "--shapes", "2", "3", "8"]),
    "t_noepi": ("conv3d_wino3", 1, {}, ["--only", "fwd", "--shapes", "2", "3", "8"]),
    "t_nostage": ("conv3d_wino3", 2, {}, ["--only", "fwd", "--shapes", "2", "3", "8"]),
    "t_noweights": ("conv3d_wino3", 4, {}, ["--only", "fwd", "--shapes", "2", "3", "8"]),
    "t_nobarrier": ("conv3d_wino3", 8, {}, ["--only", "fwd", "--shapes", "2", "3", "8"]),
    "t_bare": ("conv3d_wino3", 1 + 2 + 4 + 8, {}, ["--only", "fwd", "--shapes", "2", "3", "8"]),
    "t_bare_noreads": ("conv3d_wino3", 1 + 2 + 4 + 8 + 32, {}, ["--only", "fwd", "--shapes", "2", "3", "8"]),
    "q_stag25": ("conv3d_wino2p", 0, {"PULPO_W2P_STAGGER": "25"}, ["--only", "fwd", "--shapes", "0", "3", "8"]),
    "q_stag50": ("conv3d_wino2p", 0, {"PULPO_W2P_STAGGER": "50"}, ["--only", "fwd", "--shapes", "0", "3", "8"]),
    "q_stag75": ("conv3d_wino2p", 0, {"PULPO_W2P_STAGGER": "75"}, ["--only", "fwd", "--shapes", "0", "3", "8"]),

}


def build(only=()):
    build_library()
    os.makedirs(ABL, exist_ok=True)
    done = {}
    for name, (unit, val, _, _) in VARIANTS.items():
        if only and not any(name.startswith(o) for o in only):
            continue
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
    only = [a for a in extra if a in VARIANTS or a.endswith("*")]
    extra = [a for a in extra if a not in only]
    for name, (unit, val, env, args) in VARIANTS.items():
        if only and not any(name == o or (o.endswith("*") and name.startswith(o[:-1])) for o in only):
            continue
        lib = os.path.join(ABL, f"lib_{unit}_abl{val}.so")
        e = dict(os.environ, PULPO_HIP_LIB=lib, **env)
        print(f"==== {name}  ({os.path.basename(lib)} {env})", flush=True)
        if args is None:          # stamp build: scripts/stamps.py (stamps2.py for the pipelined kernel) on three layer shapes
            script = "stamps2.py" if unit == "conv3d_wino2p" else "stamps.py"
            for shape in (["32", "32", "160"], ["96", "96", "80"], ["64", "64", "80"]):
                subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script)] + shape, env=e, check=False)
            continue
        subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "conv_bench.py")] + args + extra, env=e, check=False)
        sys.stdout.flush()


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run(sys.argv[2:])
