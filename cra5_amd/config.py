"""Runtime configuration of the MI355X path: ONE dataclass, read from the environment in ONE place.

Everything a deployment may want to set is a field of `RuntimeConfig`; `RuntimeConfig.from_env()` is the only code of the
package that turns `CRA5_*` variables into settings ("model config as a dataclass"; VERDICT r5 item
8).  `VAEformer(..., runtime=cfg)` / `cra5_api(..., runtime=cfg)` take an explicit object; without one they call
`from_env()`.  bench.py prints `describe()` into its JSON line.

Not settings, and therefore not here:
  * the ARCHITECTURE (`cra5_amd.vaeformer.config_for`: the reference's ddconfig / priorconfig dictionaries);
  * what the launcher owns (RANK / WORLD_SIZE / MASTER_*, HIP_VISIBLE_DEVICES);
  * bit-identical implementation alternatives kept for tests (`VAEformer.compact_records`, `.fused_unembed`,
    `.resolve_on_gpu`, `.attn_balanced`, `.f16_layout`): plain attributes a test flips, no environment variable;
  * test hooks (CRA5_TEST_*, CRA5_SHARE_GPU, CRA5_FORCE_DIST, CRA5_DIST_BACKEND) and CRA5_LIB (which library file to
    load: a build matter, cra5_amd/_lib.py).
The native library reads NO environment variable (experiment overrides exist only in variant builds:
tools/build_variant.sh -DCRA5_TUNING_ENV).
"""
import dataclasses
import os
from dataclasses import dataclass


def _flag(v):
    return str(v).strip().lower() not in ("0", "false", "off", "no", "")


@dataclass
class RuntimeConfig:
    # ---- numerics ---------------------------------------------------------------------------------------------------
    precision: str = "fp32"          # "fp32": 3 x f16-split MFMA, fp32-accurate (the headline) | "f16": BASELINE configs[4]
    gemm_engine: str = "split"       # "split" (f16 matrix cores) | "f32" (exact-f32 MFMA chain: the range guard's engine)
    attn_engine: str = "split"       # the same for attention
    range_guard: bool = True         # re-run a frame whose activations left the f16 range on the exact-f32 engines
    # ---- scheduling of the frame pipeline ------------------------------------------------------------------------------
    gpu_exclusive: bool = True       # one frame's kernels at a time (single-frame API calls); the batch paths / bench overlap
    gpu_slots: int = 3               # frames inside a GPU phase at a time when phases overlap (0 = unlimited)
    inflight: int = 12               # frames in flight per GPU (bench.py, batch API default)
    switch_interval_s: float = 0.0002   # CPython GIL hand-over interval while frame threads run (0 = leave it)
    # ---- host link --------------------------------------------------------------------------------------------------
    copy_threads: int = 8            # host threads of ONE staged single-frame copy (cra5_copy_*_staged)
    batch_copy_threads: int = 1      # ... per frame of the batch API (1: one numpy copy on the frame thread)
    link_serial: bool = True         # batch API: one frame per direction on the host link at a time
    numa_bind_single: bool = True    # a 1-rank job binds itself to its GPU's NUMA node like the ranks of an N-rank job
    # ---- assets ---------------------------------------------------------------------------------------------------
    weights: str = ""                # checkpoint file or directory (zoo.vaeformer_pretrained), "" = the reference's default lookup

    ENV = {   # field -> environment variable (the ONLY place these names are read)
        "precision": "CRA5_PRECISION", "gemm_engine": "CRA5_GEMM", "attn_engine": "CRA5_ATTN", "range_guard": "CRA5_RANGE_GUARD",
        "gpu_exclusive": "CRA5_GPU_EXCLUSIVE", "gpu_slots": "CRA5_GPU_SLOTS", "inflight": "CRA5_INFLIGHT",
        "switch_interval_s": "CRA5_SWITCH_INTERVAL", "copy_threads": "CRA5_COPY_THREADS",
        "batch_copy_threads": "CRA5_BATCH_COPY_THREADS", "link_serial": "CRA5_LINK_SERIAL",
        "numa_bind_single": "CRA5_NUMA_BIND_SINGLE", "weights": "CRA5_WEIGHTS",
    }

    def __post_init__(self):
        if self.precision not in ("fp32", "f16"):
            raise ValueError("precision (CRA5_PRECISION) must be 'fp32' or 'f16'")
        if self.gemm_engine not in ("split", "f32"):
            raise ValueError("gemm_engine (CRA5_GEMM) must be 'split' or 'f32'")
        if self.attn_engine not in ("split", "f32"):
            raise ValueError("attn_engine (CRA5_ATTN) must be 'split' or 'f32'")
        if self.gpu_slots < 0 or self.inflight < 1 or not (1 <= self.copy_threads <= 64) or not (1 <= self.batch_copy_threads <= 64):
            raise ValueError("gpu_slots >= 0, inflight >= 1, 1 <= copy threads <= 64")

    @classmethod
    def from_env(cls, env=None, **overrides):
        """The defaults, overridden by the CRA5_* variables of `env` (os.environ), overridden by keyword arguments."""
        env = os.environ if env is None else env
        kw = {}
        for f in dataclasses.fields(cls):
            var = cls.ENV.get(f.name)
            if var is None or var not in env:
                continue
            raw = env[var]
            kw[f.name] = _flag(raw) if f.type in (bool, "bool") else (int(raw) if f.type in (int, "int") else (
                float(raw) if f.type in (float, "float") else raw))
        kw.update(overrides)
        return cls(**kw)

    def replace(self, **kw):
        return dataclasses.replace(self, **kw)

    def describe(self):
        """For logs / bench.py's JSON line: every field, and which of them the environment set."""
        d = dataclasses.asdict(self)
        d["set_by_environment"] = sorted(v for k, v in self.ENV.items() if v in os.environ)
        return d

