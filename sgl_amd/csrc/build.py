"""Build libsgl_hip.so (the C-ABI HIP library) and libsgl_probe.so (measurement / test support) in-tree with hipcc for gfx950.

    python -m sgl_amd.csrc.build [--force] [--verbose]

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels to the GPU box with the repo snapshot."""
import argparse
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, "libsgl_hip.so")
SOURCES = ["sgl_core.cpp", "sgl_spmm.hip", "sgl_aggregate.hip", "sgl_normalize.hip", "sgl_ingest.hip", "sgl_shims.hip", "sgl_exchange.hip", "sgl_reorder.hip"]
# measurement / test support, a library of its own (include/sgl_probe.h): memory probes, placed allocations, synthetic workloads
PROBE_LIB = os.path.join(HERE, "libsgl_probe.so")
PROBE_SOURCES = ["sgl_probe_core.cpp", "sgl_probe.hip", "sgl_synth.hip", "sgl_mem.hip"]
HEADERS = ["sgl_common.h", os.path.join(ROOT, "include", "sgl_hip.h"), os.path.join(ROOT, "include", "sgl_probe.h")]
ARCH = "gfx950"
FLAGS = [
    f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
    # numerics contract: no implicit FMA contraction (every fma in the kernels is written out, so the
    # accumulation order/rounding is exactly what the source says), IEEE division and sqrt
    "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-Wall", "-Wno-unused-function", "-Wno-unused-result",
]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (need the ROCm toolchain to build libsgl_hip.so)")
    return exe


def stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """both libraries: libsgl_hip.so (the product) and libsgl_probe.so (measurement / test support); returns the product's path"""
    hdrs = [h if os.path.isabs(h) else os.path.join(HERE, h) for h in HEADERS]
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    cc = hipcc()

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if force or stale(obj, [src] + hdrs + [os.path.abspath(__file__)]):
            cmd = [cc] + FLAGS + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
            if verbose and r.stderr.strip():
                print(r.stderr)
        return obj

    groups = ((LIB, SOURCES), (PROBE_LIB, PROBE_SOURCES))
    every = [os.path.join(HERE, s_) for _, names in groups for s_ in names]
    with ThreadPoolExecutor(max_workers=min(len(every), os.cpu_count() or 4)) as ex:
        objs = dict(zip(every, ex.map(compile_one, every)))
    for lib, names in groups:
        mine = [objs[os.path.join(HERE, s_)] for s_ in names]
        if force or stale(lib, mine):
            cmd = [cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib] + mine + ["-ldl", "-pthread"]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    print(build(a.force, a.verbose))
    sys.exit(0)
