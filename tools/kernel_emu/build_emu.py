"""Build tools/kernel_emu/libsbk_emu.so: the SAME kernel sources as the product
library, compiled with g++ against the host-side sbk_device.h of this directory.
Test tooling only (see sbk_device.h in this directory)."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "speechbrain_amd", "csrc")
LIB = os.path.join(HERE, "libsbk_emu.so")


def build(force=False, verbose=False):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip"))) + sorted(glob.glob(os.path.join(CSRC, "*.cpp")))
    srcs.append(os.path.join(HERE, "sbk_emu_runtime.cpp"))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "*.h")) + [
        os.path.join(ROOT, "include", "sbk.h")]
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    newest_h = max(os.path.getmtime(h) for h in hdrs)
    objs, procs = [], []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), newest_h):
            cmd = ["g++", "-std=c++17", "-O2", "-g", "-fPIC", "-x", "c++", "-c", s, "-o", o, "-I", HERE, "-I", CSRC,
                   "-I", os.path.join(ROOT, "include"), "-Wno-unknown-pragmas", "-fno-omit-frame-pointer"]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"g++ failed on {s}")
    if force or procs or not os.path.exists(LIB):
        subprocess.check_call(["g++", "-shared", "-fPIC", "-o", LIB] + objs + ["-lpthread"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
