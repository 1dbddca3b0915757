"""Build libetch_hip.so (HIP kernels + C ABI) for gfx950 with hipcc.  Cross-compiles without a GPU.

    python -m etch_amd.build [--force]

Per-file flags: the index kernels are built with -ffp-contract=off ("bit-defined fp32" distances).
The library is written in-tree (etch_amd/lib/) so it travels to the GPU box with the snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj_exp" if os.environ.get("ETCH_BUILD_EXPERIMENTS", "0") == "1" else "obj")
LIB = os.path.join(LIBDIR, "libetch_hip.so")
ARCH = "gfx950"
BASE = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-I", CSRC,
        "-I", os.path.join(os.path.dirname(HERE), "include")]
EXTRA = {"index_ops.hip": ["-ffp-contract=off"]}


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


EXPERIMENTS = os.environ.get("ETCH_BUILD_EXPERIMENTS", "0") == "1"
EXPERIMENT_SOURCES = {"so3conv32.hip"}         # opt-in kernels measured slower than the default path (+ the `#ifdef ETCH_BUILD_EXPERIMENTS` parts of pt.hip)


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip") and (EXPERIMENTS or f not in EXPERIMENT_SOURCES))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "etch_hip.h"))
    cc = hipcc()
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [os.path.join(CSRC, src)] + headers + [os.path.abspath(__file__)]):
            jobs.append([cc] + BASE + (["-DETCH_BUILD_EXPERIMENTS"] if EXPERIMENTS else []) + EXTRA.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    # both variants (default / ETCH_BUILD_EXPERIMENTS=1) link to the same libetch_hip.so from their own object directories: the variant last linked is
    # recorded next to the library, and a request for the other one relinks even when no source changed
    stamp = LIB + ".variant"
    variant = "experiments" if EXPERIMENTS else "default"
    linked = open(stamp).read().strip() if os.path.exists(stamp) else None
    if force or jobs or _stale(LIB, objs) or linked != variant:
        run([cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs)
        with open(stamp, "w") as f:
            f.write(variant + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
