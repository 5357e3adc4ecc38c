"""Build the in-tree native library `cra5_amd/libcra5_amd.so` for gfx950.

    python -m cra5_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with the
repo snapshot to the GPU box (it is NOT in .gpurunignore).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcra5_amd.so")
SOURCES = ["host_entropy.cpp", "gemm_f32.hip", "gemm_split_f16.hip", "attention_f32.hip", "attention_split_f16.hip", "elementwise.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(HERE, "..", "include", "cra5_amd.h"),
                                                       os.path.join(CSRC, "split.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, os.path.splitext(s)[0] + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(
                os.path.getmtime(src), os.path.getmtime(os.path.join(HERE, "..", "include", "cra5_amd.h")),
                os.path.getmtime(os.path.join(CSRC, "split.h"))):
            cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", obj]
            if s.endswith(".cpp"):
                cmd.insert(1, "-x")
                cmd.insert(2, "c++")
                cmd.remove("--offload-arch=gfx950")
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
