"""Builds libmorna_hip.so in-tree with hipcc for gfx950 (MI355X) only.

    python -m morna_amd.build [--force]

The .so is git-ignored but travels with the working tree to the GPU box.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmorna_hip.so")

FLAGS = [
    "--offload-arch=gfx950",
    "-O3", "-std=c++17", "-fPIC", "-shared",
    # a*b+c stays two roundings unless written as fmaf: the forest and feature
    # kernels are specified operation by operation (DESIGN.md "Numerics")
    "-ffp-contract=off",
    "-fno-fast-math",
    "-Wall", "-Wno-unused-function",
]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def deps():
    return sources() + glob.glob(os.path.join(CSRC, "*.hpp")) + [
        os.path.join(HERE, "..", "include", "morna_hip.h"), os.path.abspath(__file__)]


def up_to_date():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    return all(os.path.getmtime(d) <= t for d in deps())


def build(force=False, verbose=False):
    if not force and up_to_date():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if (not force and os.path.exists(obj)
                and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in deps() if not d.endswith(".hip"))
                and os.path.getmtime(src) <= os.path.getmtime(obj)):
            continue
        cmd = [hipcc] + [f for f in FLAGS if f != "-shared"] + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lz", "-ldl"]   # zlib: parse.hip; dl: RCCL is looked up at run time (comm.hip)
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
