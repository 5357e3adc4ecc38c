"""Build the in-tree native library `cra5_amd/libcra5_amd.so` for gfx950.

    python -m cra5_amd.build [--force] [--flavour release|debug|rangecheck|asan]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with the
repo snapshot to the GPU box (it is NOT in .gpurunignore).

Flavours (the reference's setup.py:72-75 has a debug switch with `-O0 -g -UNDEBUG`):
  release     -O3                                   -> cra5_amd/libcra5_amd.so   (the product)
  debug       -O0 -g -UNDEBUG (asserts on; host code; device code -O1 -g: -O0 kernels spill
              past the 64 KB scratch limit)         -> cra5_amd/_flavours/libcra5_debug.so
  rangecheck  -O3 -DCRA5_RANGE_CHECK: counts |x| >= 65504 / non-finite values in every
              split-f16 producer (cra5_debug_range_counts) -> cra5_amd/_flavours/libcra5_rangecheck.so
  asan        host entropy coder with -fsanitize=address,undefined (clang shared runtime); run with
              LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so) -> cra5_amd/_flavours/libcra5_asan.so
Select a non-release flavour at run time with CRA5_LIB=<path>.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcra5_amd.so")
SOURCES = ["host_entropy.cpp", "gemm_f32.hip", "gemm_split_f16.hip", "attention_f32.hip", "attention_split_f16.hip",
           "elementwise.hip", "hyper.hip", "runtime.hip"]
SOURCES = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
HEADERS = [os.path.join(ROOT, "include", "cra5_amd.h"), os.path.join(CSRC, "split.h"),
           os.path.join(CSRC, "gemm_split_epilogue.inc"), os.path.join(CSRC, "gemm_split_epilogue_fast.inc"),
           os.path.join(CSRC, "gemm_split_epilogue_unembed.inc")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# per-file flags.  gemm_split_f16.hip: the epilogue's row-block loop must unroll FULLY in every instantiation (a
# dynamically indexed accumulator array lives in scratch); the 128 x 128-per-wave one exceeds clang's default
# 16 K-instruction budget for `#pragma unroll`.
# -Wno-inline-asm: the hand-written LDS-DMA names m0 as a clobber (it is the DMA's LDS base by definition).
EXTRA = {"gemm_split_f16.hip": ["-mllvm", "-pragma-unroll-threshold=262144", "-Wno-inline-asm"]}

FLAVOURS = {
    "release": dict(host=["-O3"], dev=["-O3"], link=[]),
    "debug": dict(host=["-O0", "-g", "-UNDEBUG"], dev=["-O1", "-g", "-UNDEBUG"], link=[]),
    "rangecheck": dict(host=["-O3"], dev=["-O3", "-DCRA5_RANGE_CHECK"], link=[]),
    "asan": dict(host=["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-shared-libsan"],
                 dev=["-O3"], link=["-fsanitize=address,undefined", "-shared-libsan"]),
}


def lib_path(flavour="release"):
    if flavour == "release":
        return LIB
    return os.path.join(HERE, "_flavours", f"libcra5_{flavour}.so")   # travels to the GPU box (objects do not)


def _newer(path, deps):
    return not os.path.exists(path) or any(os.path.getmtime(d) > os.path.getmtime(path) for d in deps)


def build(force=False, verbose=False, flavour="release"):
    fl = FLAVOURS[flavour]
    out = lib_path(flavour)
    objdir = CSRC if flavour == "release" else os.path.join(ROOT, "build_variants", flavour)
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    if not force and not _newer(out, srcs + HEADERS):
        return out
    objs = []
    if flavour != "release" and fl["dev"] == FLAVOURS["release"]["dev"]:
        build(force=False, verbose=verbose)   # device objects are the release ones: reuse them
    for s, src in zip(SOURCES, srcs):
        obj = os.path.join(objdir, os.path.splitext(s)[0] + ".o")
        if flavour != "release" and not s.endswith(".cpp") and fl["dev"] == FLAVOURS["release"]["dev"]:
            objs.append(os.path.join(CSRC, os.path.splitext(s)[0] + ".o"))
            continue
        if force or _newer(obj, [src] + HEADERS):
            if s.endswith(".cpp"):
                cmd = [HIPCC, "-x", "c++", "-std=c++17", "-fPIC"] + fl["host"] + ["-c", src, "-o", obj]
            else:
                cmd = [HIPCC, "--offload-arch=gfx950", "-std=c++17", "-fPIC"] + fl["dev"] + EXTRA.get(s, []) + \
                      ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + fl["link"] + ["-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    fl = "release"
    if "--flavour" in sys.argv:
        fl = sys.argv[sys.argv.index("--flavour") + 1]
    print(build(force="--force" in sys.argv, verbose=True, flavour=fl))
