"""Kernels of libetch_hip.so that use scratch memory (register spills / private arrays), from the code objects' metadata.
   python profiles/scripts/spills.py   (cross-compiles every csrc/*.hip for gfx950 to assembly; no GPU needed)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "etch_amd", "csrc")
def demangle(n):
    for tool in ("c++filt", "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"):
        try:
            return subprocess.run([tool, n], capture_output=True, text=True).stdout.strip() or n
        except FileNotFoundError:
            continue
    return n
rows = []
with tempfile.TemporaryDirectory() as tmp:
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith(".hip") or f == "so3conv32.hip":
            continue
        out = os.path.join(tmp, f + ".s")
        extra = ["-ffp-contract=off"] if f == "index_ops.hip" else []
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-S",
                        "--cuda-device-only", "-o", out, os.path.join(CSRC, f)] + extra, check=True, capture_output=True)
        txt = open(out).read()
        for b in txt.split("  - .agpr_count:")[1:]:
            g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, b).group(1))
            name = re.search(r"\.name:\s+(\S+)", b).group(1)
            rows.append((f, demangle(name), g("private_segment_fixed_size"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("vgpr_count")))
print("%d kernels; with scratch:" % len(rows))
for f, n, priv, vs, ss, vg in rows:
    if priv or vs:
        print("  %-16s %-100s scratch %4d B  vgpr spills %3d  vgprs %3d" % (f, n[:100], priv, vs, vg))
