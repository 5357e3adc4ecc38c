"""The C-ABI shared library loads without a GPU and exports every symbol that
include/cra5_amd.h declares (no device compute is called here)."""
import ctypes
import os
import re

from cra5_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "cra5_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cra5_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_table_agree():
    assert declared_symbols() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "run `python -m cra5_amd.build`"
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(L, name), name
    assert _lib.lib().cra5_abi_version() == 1


def test_argument_validation_without_gpu():
    """Launchers validate arguments before touching the device: bad calls return
    CRA5_ERR_ARG (-7) even on a box with no GPU."""
    L = _lib.lib()
    assert L.cra5_gemm_nt_f32(None, 0, None, 0, None, 0, None, None, 0, 1, 1, 4, 0, None) == -7
    assert L.cra5_layernorm_f32(None, 0, None, None, None, 0, None, 0, 1, 3, 1e-6, 0, None) == -7
    assert L.cra5_window_attention_f32(None, None, None, None, 0, 64, 1, 1, 1, 1, 1, 1.0, None) == -7
    assert L.cra5_gemm_nt_split(None, 32, None, 32, None, 0, None, 0, None, None, 0, 1, 1, 32, 1.0, 0, None) == -7
    assert L.cra5_split_f16(None, 0, None, 1, 1, 32, 1.0, None) == -7
    assert L.cra5_window_attention_split(None, 0, None, None, None, 0, 64, 1, 1, 1, 1, 1, 1.0, 0, None) == -7
    assert L.cra5_pmf_to_quantized_cdf(None, 0, 16, None) == -7
    assert L.cra5_small_gemm_nt_split(None, 32, None, 32, None, 0, None, 0, None, None, 0, 1, 1, 32, 1.0, 0, 0, 0, 0, 0,
                                      None) == -7
    assert L.cra5_hyper_attention_f32(None, None, None, 0, 1, 72, 1, 1.0, None) == -7
    assert L.cra5_conv_im2col_f32(None, None, 1, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1, 32, None) == -7
    assert L.cra5_deconv_col2im_f32(None, None, None, 1, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1, 1, None) == -7
    assert L.cra5_unary_f32(None, None, 1, 0, 0.0, None) == -7
    assert L.cra5_debug_range_counts(None, 0) == -8          # not compiled into the release flavour


def test_no_cuda_shims_or_dual_paths_in_sources():
    csrc = os.path.join(ROOT, "cra5_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".cpp", ".h")):
            s = open(os.path.join(csrc, f)).read()
            assert "__HIP_PLATFORM_AMD__" not in s and "cuda_runtime" not in s and "hipify" not in s.lower()


def test_release_sources_carry_no_experiment_switches():
    """the production translation units once carried ~30 experiment macros (GEMM_PF, GEMM_SKIP_*,
    ATT_SKIP_*, ...), several of which build a kernel that is wrong by design - one stray -D away from shipping.
    The losers are in the history (git show fa7b4dd:tools/probes/archive/).  Every preprocessor conditional left in cra5_amd/csrc must
    test one of the macros below, and the build recipe must define none of them for the release flavour."""
    import re as _re
    allowed = {"__HIP_DEVICE_COMPILE__",     # host / device pass of hipcc
               "__x86_64__",
               "CRA5_RANGE_CHECK",           # `rangecheck` flavour: counts out-of-range split-f16 stores (same results)
               "CRA5_GEMM_TRACE",            # tools/gemm_trace.py: per-work-group timestamps (same results)
               "CRA5_ATTN_TRACE",            # tools/attn_trace.py: per-work-group / per-unit timestamps of the windowed attention
               "CRA5_HY_SWEEP",              # tools/hyper_gemm_sweep.sh: CRA5_HY_GEMM override of the small-GEMM shape
               "CRA5_TUNING_ENV"}            # variant builds: CRA5_GEMM_TILE / CRA5_ATT72_NW tile overrides (same results)
    csrc = os.path.join(ROOT, "cra5_amd", "csrc")
    seen = set()
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".hip", ".cpp", ".h", ".inc")):
            continue
        for line in open(os.path.join(csrc, f)):
            m = _re.match(r"\s*#\s*(if|ifdef|ifndef|elif)\b(.*)", line)
            if not m:
                continue
            names = set(_re.findall(r"[A-Za-z_][A-Za-z0-9_]*", m.group(2))) - {"defined"}
            names = {n for n in names if not n.isdigit()}
            assert names <= allowed, (f, line.strip())
            seen |= names
    from cra5_amd import build as B
    flags = " ".join(B.FLAVOURS["release"]["host"] + B.FLAVOURS["release"]["dev"] + sum(B.EXTRA.values(), []))
    assert "-D" not in flags, flags


def test_product_library_reads_no_environment_variable():
    """Round 6: settings come from ONE place, cra5_amd/config.py RuntimeConfig.from_env(); the native
    library itself imports no getenv (the tile-override hooks of earlier rounds exist in -DCRA5_TUNING_ENV variant builds
    only) and no other module of the package reads a CRA5_* setting from os.environ."""
    import subprocess
    from cra5_amd import build as B
    lib = B.build()
    und = subprocess.run(["nm", "-D", "--undefined-only", lib], capture_output=True, text=True).stdout
    assert "getenv" not in und, [l for l in und.split("\n") if "getenv" in l]
    import re as _re
    pkg = os.path.join(ROOT, "cra5_amd")
    allowed_elsewhere = {"CRA5_LIB",                                            # which library file: build matter (_lib.py)
                         "CRA5_FORCE_DIST", "CRA5_DIST_BACKEND", "CRA5_SHARE_GPU", "CRA5_DIST_TIMEOUT_S",   # launcher / tests (dist.py)
                         "CRA5_TEST_FREE_GIB", "CRA5_TEST_N_CPUS"}              # test hooks of the preflight
    for f in sorted(os.listdir(pkg)):
        if not f.endswith(".py") or f in ("config.py", "build.py"):
            continue
        src = open(os.path.join(pkg, f)).read()
        for m in _re.finditer(r"environ[^\n]*?[\"'](CRA5_[A-Z0-9_]+)[\"']", src):
            assert m.group(1) in allowed_elsewhere, (f, m.group(1))
