"""The ONE place the package reads its environment: every run-time switch of unidefense_amd is a field of `cfg`,
filled once at import from the `UD_*` variables listed here and changeable afterwards from Python
(`from unidefense_amd.config import cfg; cfg.deterministic = True`) — the modules look the field up when they act,
not when they are imported, EXCEPT the four marked (import): gemm_tune_defaults, the load side of gemm_tune_cache and
lib_path are consumed when unidefense_amd.kernels / .lib are imported, engine_graph when an engine is built — set those
through the environment (tools/run_with.py does).  Anything not listed here is a compile-time constant of the kernels.

| field | env | default | meaning |
|---|---|---|---|
| deterministic | UD_DETERMINISTIC | 0 | split-K GEMMs add their partial products in a FIXED order (slices + ud_sum_slices) instead of fp32 atomics: the step no longer depends on the ORDER in which fp32 atomics land (what remains order-dependent are the fp64 atomic sums of the fused MBConv path, at 1e-16: repeatable in every run observed, tests/test_y_fullsize_gpu.py, not guaranteed bitwise; the operator path, cfg.fused_mbconv = False, reduces in a fixed order throughout) (measured: +2.3 ms on the 33.5 ms bs-32 step, which is why it is opt-in; the parity suite runs with it, tests/conftest.py) |
| fused_mbconv | UD_FUSED_MBCONV | 1 | training-mode MBConv blocks as one tape node with deferred BatchNorm (0: operator by operator) |
| half_storage | UD_HALF_STORAGE | 0 | fp16 activation storage in the MBConv trunk (BASELINE configs[4]); `model.half_storage` overrides |
| gemm_tune | UD_GEMM_TUNE | 1 | measure (tile, split-K) candidates on the first eager call of an unseen GEMM shape |
| gemm_tune_defaults (import) | UD_GEMM_TUNE_DEFAULTS | 1 | start from the shipped plans (gemm_plans_gfx950.json) |
| gemm_tune_cache | UD_GEMM_TUNE_CACHE | unset | JSON file the measured plans are added to (and read from at import) |
| engine_graph (engine init) | UD_ENGINE_GRAPH | 1 | the engine's two passes replayed from hipGraphs |
| syncbn_exchange | UD_SYNCBN_EXCHANGE | 1 | SyncBN sums through the peer-mapped mailbox kernel (0: dist.all_reduce) |
| force_collectives | UD_FORCE_COLLECTIVES | 0 | issue the data-parallel collectives even in a world of one rank (single-GPU test of the RCCL path) |
| hip_adamw | UD_HIP_ADAMW | 1 | build_optimizer returns the multi-tensor HIP AdamW for 'adamw' on the GPU |
| lib_path (import) | UD_LIB_PATH | unset | load another build of libunidefense_hip.so (A/B of kernel builds) |
| spectral_p2 | UD_SPECTRAL_P2 | auto | the spectral 1x1 convs' forward / data-gradient GEMMs from pre-split fp16 x 2 planes (ud_gemm_p3 prec 2): `auto` where measured (or, untuned, estimated) faster than the in-kernel bf16 x 3 split, `on` wherever the kernel takes the shape, `off` never |
| weight_plane_batch | UD_WEIGHT_PLANE_BATCH | 1 | the planes of all conv weights on the planes path made by two launches at the start of a forward (ud_split_planes_h2t_multi) instead of two per weight; a weight joins on its first eager use, planes are handed out only while the parameter's version is the one they were made from |
| side_branch | UD_SIDE_BRANCH | 0 | training forward / backward of UDEB4: the reconstruction decoder and its loss tail on a second stream beside the trunk's stage 5 ... head (tape.side_branch).  Correct (the model and engine goldens pass with it) but measured SLOWER on this runtime: 24.07 -> 24.44 ms per step, the two hardware queues overlap for 0.25 of the branch's 1.85 ms (profiles/r06/side_branch_ab.txt) |
| expand_bwd_fused | UD_EXPAND_BWD_FUSED | 1 | the thin expand convs' backward (24 -> 144, 32 -> 192: the 128 x 128 / 64 x 64 blocks) as ONE pass over (dz, e): BatchNorm backward applied on load, weight and data gradient from one LDS image (ud_pw_bwd_fused) instead of ud_normbwd_apply + two gemm_x3 launches |
| project_bwd_fused | UD_PROJECT_BWD_FUSED | 1 | the thin project convs' backward (144 / 192 -> 32: the 64 x 64 blocks) without the conv's data gradient in HBM: weight gradient + SE dot in one pass over d, gate / swish backward + BatchNorm-1 sums in a second, each re-making its 32-row tile of dc = dp Wp from the thin dp (ud_pj_bwd_fused_a / _b) instead of two gemm_x3 launches + ud_coldot_bn + ud_se_scale_bwd_bn; and their forward in one pass over d (ud_pj_fwd_fused: gate applied on load, BatchNorm-2 statistics out of the epilogue, the gated tensor never written) instead of ud_se_scale_bn + gemm_x3 |
| project_bwd_fused_wide | UD_PROJECT_BWD_FUSED_WIDE | 1 | ... also for the 32 x 32 blocks' project convs (192 / 336 -> 56), their tensors walked as 96- / 112-channel column chunks (0: only the 64 x 64 blocks) |
| project_fused_narrow | UD_PROJECT_FUSED_NARROW | 0 | ... and for the 128 x 128 blocks' project convs (48 / 24 -> 24).  Built and operator-tested (tests/test_b_fused_kernels_gpu.py), worth 0.2 ms of the bs-32 step, and OFF: with it the N = 8 fixture's blocks.9 gate gradient — a single heavily cancelling sum on which the oracle's own fp32 run is 1.8e-5 off float64 — moves from 0.19 to 0.74 of the plain bound in the `bench` run mode and 1 % past the reference's fp32 record (profiles/r06/pj_bwd_fused.txt); not a margin to ship on |

The shared library itself reads four variables when it is loaded, for hosts that do not go through Python:
UD_GEMM_PATH (the initial `ud_gemm_set_path` value: 0 auto, 1 fp32 pipe, 2 split-bf16 everywhere, 3 fp16 MFMA), UD_FFT32_WAVE
(the initial `ud_fft32_set_wave` form of the 32 x 32 transforms: 0 one lane per row, 1 lane pairs; unset: per shape) and the
kernel-bench overrides UD_GEMM_CFG / UD_GEMM_X3_CFG (force one tile configuration; tools/bench_gemm.py).
"""
import os
from dataclasses import dataclass, fields
from typing import Optional


def _flag(name, default):
    return os.environ.get(name, "1" if default else "0") == "1"


@dataclass
class Config:
    deterministic: bool = False
    fused_mbconv: bool = True
    half_storage: bool = False
    gemm_tune: bool = True
    gemm_tune_defaults: bool = True
    gemm_tune_cache: Optional[str] = None
    engine_graph: bool = True
    syncbn_exchange: bool = True
    force_collectives: bool = False
    hip_adamw: bool = True
    lib_path: Optional[str] = None
    spectral_p2: str = "auto"
    weight_plane_batch: bool = True
    side_branch: bool = False
    expand_bwd_fused: bool = True
    project_bwd_fused: bool = True
    project_bwd_fused_wide: bool = True
    project_fused_narrow: bool = False

    @classmethod
    def from_env(cls):
        c = cls()
        for f in fields(cls):
            env = "UD_" + f.name.upper()
            if f.type is bool:
                setattr(c, f.name, _flag(env, f.default))
            else:
                setattr(c, f.name, os.environ.get(env) or f.default)
        if c.spectral_p2 not in ("auto", "on", "off"):
            raise ValueError(f"UD_SPECTRAL_P2 must be auto, on or off, got {c.spectral_p2!r}")
        return c

    def describe(self):
        return {f.name: getattr(self, f.name) for f in fields(self)}


cfg = Config.from_env()


class override:
    """`with override(deterministic=False): ...` — tests and A/B tools."""

    def __init__(self, **kw):
        self.kw, self.saved = kw, {}

    def __enter__(self):
        for k, v in self.kw.items():
            if not hasattr(cfg, k):
                raise AttributeError(k)
            self.saved[k] = getattr(cfg, k)
            setattr(cfg, k, v)
        return cfg

    def __exit__(self, *exc):
        for k, v in self.saved.items():
            setattr(cfg, k, v)
        return False
