"""In-tree build of libgpe.so (the C-ABI of include/gpe.h) with hipcc for gfx950.

hipcc cross-compiles without a GPU; the resulting .so travels to the GPU box with the snapshot.
Flags that matter for parity: -ffp-contract=off (no FMA contraction; the oracle is compiled the
same way) and hipcc's default correctly-rounded f32 divide/sqrt (never -ffast-math).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(HERE, "csrc", "build")
LIB = os.path.join(HERE, "libgpe.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

CXXFLAGS = [
    "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-ffp-contract=off",
    "-fno-fast-math", "-Wall", "-Wno-unused-function",
] + os.environ.get("GPE_EXTRA_CXXFLAGS", "").split()     # e.g. -DGPE_TILE_STAMPS for the diagnostic build


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "gpe.h"))
    return hs


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(BUILD, os.path.basename(src)[:-4] + ".o")
    if _stale(obj, [src] + _headers()):
        cmd = [HIPCC] + CXXFLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build_library(force=False, jobs=4):
    os.makedirs(BUILD, exist_ok=True)
    if force:
        for f in os.listdir(BUILD):
            os.remove(os.path.join(BUILD, f))
        if os.path.exists(LIB):
            os.remove(LIB)
    srcs = sources()
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(_compile, srcs))
    if _stale(LIB, objs):
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
