"""Builds the HIP library in-tree: gnnflow_amd/csrc/libgnnflow_hip.so (gfx950).

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the
GPU box with the gpurun snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libgnnflow_hip.so")
SOURCES = ["capi.hip", "edge_store.hip", "sampler.hip", "feature_cache.hip", "memory_ops.hip", "block_ops.hip", "partition.hip", "ingest_sort.hip", "comm.hip"]
# every header under csrc/ (a new one is picked up without touching this file) + the public ones
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))) + [
    os.path.join("..", "..", "include", "gnnflow_hip.h"),
    os.path.join("..", "..", "include", "gnnflow_rng.h")]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "--offload-arch=" + ARCH, "-fPIC", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-result", "-ffp-contract=off"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, h) for h in HEADERS]
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(o, [s] + headers):
            jobs.append([hipcc] + FLAGS + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for warn in ex.map(run, jobs):
                if verbose and warn:
                    print(warn, file=sys.stderr)
    if force or jobs or _newer(LIB, objs):
        run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


def pybind_module_path() -> str:
    import sysconfig
    return os.path.join(CSRC, "libgnnflow" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_pybind(force: bool = False) -> str:
    """Builds the pybind11 module `libgnnflow` (the reference's native module name and
    surface, gnnflow/csrc/api.cc) over the C ABI; plain g++, links libgnnflow_hip.so."""
    import sysconfig
    import pybind11
    build()
    out = pybind_module_path()
    src = os.path.join(CSRC, "pybind_libgnnflow.cpp")
    hdr = os.path.join(CSRC, "..", "..", "include", "gnnflow_hip.h")
    if not force and not _newer(out, [src, hdr, LIB]):
        return out
    cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden",
           "-I" + sysconfig.get_paths()["include"], "-I" + pybind11.get_include(),
           src, "-o", out, "-L" + CSRC, "-lgnnflow_hip", "-Wl,-rpath,$ORIGIN"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("g++ failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_pybind(force="--force" in sys.argv))
