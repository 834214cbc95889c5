"""Build libsbk_hip.so (gfx950) from the HIP sources in this directory.

    python -m speechbrain_amd.csrc.build          # hipcc cross-compiles without a GPU

The library is built IN-TREE (speechbrain_amd/csrc/libsbk_hip.so) so that it
travels with the repo snapshot to the GPU box; it is git-ignored.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, "libsbk_hip.so")


def sources():
    return sorted(glob.glob(os.path.join(HERE, "*.hip"))) + sorted(glob.glob(os.path.join(HERE, "*.cpp")))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = sources()
    deps = srcs + glob.glob(os.path.join(HERE, "*.h")) + glob.glob(os.path.join(HERE, "hip", "*.h")) + [
        os.path.join(ROOT, "include", "sbk.h")]
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(o)
        if force or _stale(o, [s] + [d for d in deps if d.endswith(".h")]):
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c", s, "-o", o,
                   "-I", os.path.join(HERE, "hip"), "-I", HERE, "-I", os.path.join(ROOT, "include"),
                   "-Wno-unused-result"]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    if force or procs or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
