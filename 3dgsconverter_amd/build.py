"""Build libgsx_hip.so in-tree for gfx950 (hipcc cross-compiles without a GPU).

    python 3dgsconverter_amd/build.py [--force]

-ffp-contract=off: the float64 distance and the numpy-order sums must not be fused into
FMAs (HIP's __dadd_rn/__dmul_rn are plain operators and contract under the default
-ffp-contract=fast); FMAs that are wanted are written as explicit fmaf().
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libgsx_hip.so")
OBJ = os.path.join(HERE, "build")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fno-gpu-rdc", "-munsafe-fp-atomics",
         "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def newest_dep() -> float:
    t = 0.0
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in os.listdir(root):
            t = max(t, os.path.getmtime(os.path.join(root, f)))
    return max(t, os.path.getmtime(__file__))


def variant_paths(extra):
    """Experiment builds (GSX_EXTRA_FLAGS, e.g. -DGSX_ABLATE or a tuning -D) never touch the product library: they get
    their own object directory and their own output, named after the flag set -- a later plain build() cannot mistake an
    ablation library for the product.  `GSX_LIB_PATH=<that file>` makes _lib load it."""
    import hashlib
    tag = os.environ.get("GSX_VARIANT_TAG") or hashlib.sha1(" ".join(extra).encode()).hexdigest()[:10]
    return os.path.join(OBJ, "variant_" + tag), os.path.join(HERE, "variants", "libgsx_hip_%s.so" % tag)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("GSX_EXTRA_FLAGS", "").split()  # experiment builds only: separate objects + output
    OBJ, OUT = (globals()["OBJ"], globals()["OUT"]) if not extra else variant_paths(extra)
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    dep_t = newest_dep()
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= dep_t:
        return OUT

    only = set(os.environ.get("GSX_VARIANT_ONLY", "").split()) if extra else set()   # e.g. "sor_grid.hip": the other
    if only:                                                                          # objects come from the product build
        _plain_env = dict(os.environ)
        os.environ.pop("GSX_EXTRA_FLAGS")
        try:
            build(force=False, verbose=verbose)
        finally:
            os.environ.update(_plain_env)

    def compile_one(src):
        if only and src not in only:
            return os.path.join(globals()["OBJ"], src[:-4] + ".o")
        obj = os.path.join(OBJ, src[:-4] + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= dep_t:
            return obj
        cmd = [hipcc, *FLAGS, *extra, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr))
        if r.stderr.strip() and verbose:
            print(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
        objs = list(ex.map(compile_one, sources()))
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-pthread", *objs, "-o", OUT]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
