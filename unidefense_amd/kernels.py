"""Thin launch wrappers over the C ABI (unidefense_amd.lib): allocate outputs as torch tensors on the
current device and enqueue the HIP kernels on torch's current stream.  No autograd here, no math in
torch — torch only owns device memory and the stream.

Tensor convention: fp32, contiguous, pixel-major [N, H, W, C] (a row-major matrix [N*H*W, C]);
image-domain 3-channel tensors are planes [N, 3, H, W].
"""
import ctypes as C
import math
import os

import torch

from . import lib as _lib
from .config import cfg as CFG
from .lib import ConvGeom, GemmDesc, GemmP3Desc

_call = _lib.call

# when set to a list, every ud_gemm launch is bracketed by HIP events: (start, end, flops) tuples
GEMM_PROFILE = None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# Two streams (tape.side_branch: the reconstruction decoder beside the trunk): every scratch buffer / zero pool below is
# shared by the launches of ONE stream in order — a reduction's finalize launch has consumed the scratch before the next
# reduction on that stream starts — so the pools are keyed by (device, branch) and the branch's launches take their own.
_BR = 0


class branch:
    def __init__(self, b):
        self.b = b

    def __enter__(self):
        global _BR
        self.prev, _BR = _BR, self.b
        return self

    def __exit__(self, *exc):
        global _BR
        _BR = self.prev
        return False


def _key(ref):
    return (ref.device.index, _BR)


def _p(t):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError(f"expected a contiguous fp32 CUDA tensor, got {t.dtype} {t.device} "
                             f"contiguous={t.is_contiguous()} shape={tuple(t.shape)}")


def _act(*ts):
    """Activation tensors of the trunk: contiguous CUDA, all fp32 or all fp16 (half storage, BASELINE configs[4]).
    Returns the `f16` flag of the C ABI."""
    dt = None
    for t in ts:
        if t is None:
            continue
        if not (t.is_cuda and t.dtype in (torch.float32, torch.float16) and t.is_contiguous()):
            raise ValueError(f"expected a contiguous fp32 / fp16 CUDA tensor, got {t.dtype} {t.device} "
                             f"contiguous={t.is_contiguous()} shape={tuple(t.shape)}")
        if dt is not None and t.dtype != dt:
            raise ValueError(f"mixed storage types in one call: {dt} and {t.dtype}")
        dt = t.dtype
    return 1 if dt == torch.float16 else 0


def empty(shape, like, dtype=torch.float32):
    return torch.empty(shape, dtype=dtype, device=like.device)


# Zero-initialised outputs of the split-K launches (atomic accumulation): ~180 per step, each a 5 us fill launch of
# its own.  They are carved instead from a few large zero blocks (one fill per 32 MB); a block is never reused, so
# every carve is still zero.  reset_zero_pool() at the start of a forward / backward makes a step captured into a
# hipGraph contain the fills of every block it carves from.
_ZERO_POOL_ON = True
_ZERO_BLOCK = 8 << 20            # floats per block (32 MB)
_ZERO_OWN = 2 << 20              # tensors of at least this many floats get their own torch.zeros
_ZERO_POOL = {}


def reset_zero_pool():
    _ZERO_POOL.clear()
    _ZERO64_POOL.clear()


# fp64 accumulators of the fused path (BatchNorm sums, SE pooling sums, gate gradients): carved from zero blocks the
# same way — one fill per block instead of one per accumulator (~10 accumulators per MBConv block and pass).
_ZERO64_BLOCK = 1 << 20        # doubles per block (8 MB)
_ZERO64_POOL = {}


def zeros64(n, like):
    """n zero-initialised doubles (a view; never recycled within a forward / backward)."""
    n = int(n)
    key = _key(like)
    st = _ZERO64_POOL.get(key)
    capturing = torch.cuda.is_current_stream_capturing()
    if st is None or st[1] + n > st[0].numel() or st[2] != capturing:
        st = _ZERO64_POOL[key] = [torch.zeros(max(_ZERO64_BLOCK, n), dtype=torch.float64, device=like.device), 0,
                                  capturing]
    out = st[0][st[1]:st[1] + n]
    st[1] += (n + 31) // 32 * 32
    return out


def zeros(shape, like):
    n = 1
    for d in shape:
        n *= int(d)
    if not _ZERO_POOL_ON or n >= _ZERO_OWN or n == 0:
        return torch.zeros(shape, dtype=torch.float32, device=like.device)
    key = _key(like)
    st = _ZERO_POOL.get(key)
    capturing = torch.cuda.is_current_stream_capturing()
    # a block filled outside a capture must not serve carves inside one (the replay would not re-zero it), nor vice versa
    if st is None or st[1] + n > _ZERO_BLOCK or st[2] != capturing:
        st = _ZERO_POOL[key] = [torch.zeros(_ZERO_BLOCK, dtype=torch.float32, device=like.device), 0, capturing]
    out = st[0][st[1]:st[1] + n].view(shape)
    st[1] += (n + 63) // 64 * 64
    return out


def split_out(shape, like):
    """Result buffer of a split-K launch: zeros for the atomics, an uninitialised buffer marked fresh for the ordered
    slice sum (which then writes instead of adding: no fill, no read of zeros)."""
    if not CFG.deterministic:
        return zeros(shape, like)
    out = torch.empty(shape, dtype=torch.float32, device=like.device)
    out._ud_fresh = True
    return out


# ---------------------------------------------------------------------------------------------
# GEMM family
# ---------------------------------------------------------------------------------------------
# CFG.deterministic (UD_DETERMINISTIC=1): split-K GEMMs store their partial products into slices of a scratch
# buffer and add them in ascending order (ud_gemm out_mode 3 + ud_sum_slices) instead of fp32 atomics, so the same inputs
# give the same result on every run and box (the fp64 accumulators of the fused MBConv path are order-dependent at 1e-16
# only).  Default (False): fp32 atomics — one launch less per split GEMM (2.3 ms of the 35.8 ms deterministic bs-32 step),
# last-bit run-to-run variation.
_SLICE_WS = {}


def _slice_ws(ref, n):
    ws = _SLICE_WS.get(_key(ref))
    if ws is None or ws.numel() < n:
        ws = _SLICE_WS[_key(ref)] = torch.empty(max(n, 1 << 22), dtype=torch.float32, device=ref.device)
    return ws


def _gemm(A, B, Cout, M, N, K, lda, ldb, ldc, a_mode, b_mode, out_mode=0, split_k=1, geom=None,
          batch=1, strideA=0, strideB=0, strideC=0, a_off=0, b_off=0, stats=None, cfg=0):
    """stats: fp64 accumulator [sum | sumsq] (2N doubles) the epilogue should add the result's column sums into; returns
    (Cout, True) when the kernel did (ud_gemm_stats_slots), (Cout, False) when the caller still has to run colstats."""
    d = GemmDesc()
    d.A = A.data_ptr() + A.element_size() * a_off
    d.B = B.data_ptr() + B.element_size() * b_off
    d.C = Cout.data_ptr()
    d.half_mask = (A.dtype == torch.float16) | ((B.dtype == torch.float16) << 1) | ((Cout.dtype == torch.float16) << 2)
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = lda, ldb, ldc
    d.a_mode, d.b_mode, d.out_mode, d.split_k = a_mode, b_mode, out_mode, split_k
    d.batch, d.strideA, d.strideB, d.strideC = batch, strideA, strideB, strideC
    d.tile_cfg = cfg | (0x100 if _XCD_CONTIGUOUS else 0)
    slices = None
    if CFG.deterministic and out_mode == 2:
        # Cout holds a term to add to (or is a fresh buffer of split_out): result = [Cout +] the splits' partials in
        # ascending order
        total = M * ldc
        fresh = getattr(Cout, "_ud_fresh", False)
        if batch == 1 and total % 4 == 0 and Cout.dtype == torch.float32:
            stride = total
            ws = _slice_ws(Cout, split_k * stride)
            d.C, d.out_mode, d.slice_stride = ws.data_ptr(), 3, stride
            slices = (ws, stride, total, 0 if fresh else 1)
        else:                                   # no slice form for this output: one plain accumulating launch
            if fresh:
                Cout.zero_()
            d.out_mode, d.split_k = 1, 1
        if fresh:
            Cout._ud_fresh = False
    if geom is not None:
        d.g = geom
    fold = None
    if stats is not None:
        slots = _call("ud_gemm_stats_slots", C.byref(d)) if _GEMM_EPILOGUE_STATS else 0
        if slots == 0:
            stats_done = False
        else:
            stats_done = True
            tgt = stats if slots == 1 else zeros64(2 * slots * N, Cout)
            d.stat_sum, d.stat_sumsq = tgt.data_ptr(), tgt.data_ptr() + 8 * (slots * N if slots > 1 else N)
            if slots > 1:
                fold = (tgt, slots)
    if GEMM_PROFILE is not None:
        # live HIP-event timing of the dominant kernel on the stream it is launched on (bench.py roofline)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _call("ud_gemm", C.byref(d), _stream())
        e1.record()
        GEMM_PROFILE.append((e0, e1, 2.0 * M * N * K * batch, (M, N, K, a_mode, b_mode, split_k, batch),
                             _call("ud_gemm_query_path", C.byref(d)),
                             # operand + result bytes, every matrix touched once, in their storage types
                             float(batch) * (A.element_size() * M * K + B.element_size() * K * N + Cout.element_size() * M * N)))
    else:
        _call("ud_gemm", C.byref(d), _stream())
    if slices is not None:
        ws, stride, total, accumulate = slices
        _call("ud_sum_slices", _p(ws), _p(Cout), split_k, total, stride, accumulate, _stream())
    if fold is not None:
        tgt, slots = fold
        _call("ud_stat_slots_fold", _pd64(tgt), _pd64(tgt, slots * N), slots, N, _pd64(stats), _pd64(stats, N), _stream())
    return (Cout, stats_done) if stats is not None else Cout


class Planes:
    """A matrix X[R][C] as 16-bit planes in the P32 panel layout of ud_gemm_p3 (include/unidefense_hip.h): piece p of X[r][c]
    at p * plane + (c // 32) * panel + r * 32 + c % 32.  prec 3: three bf16 planes (exact split); prec 2: two fp16 planes of
    the scaled matrix + inv = 1 / scale — one scale for the tensor (scale_stride 0; readable in both GEMM modes) or one per
    row (scale_stride 1; mode 0 only)."""
    __slots__ = ("buf", "R", "C", "panel", "plane", "npanel", "prec", "inv", "scale_stride")

    def __init__(self, R, Cc, like, prec=3, per_row=False):
        self.R, self.C, self.prec = R, Cc, prec
        self.npanel = -(-Cc // 32)
        self.panel = 32 * (-(-R // 128) * 128)          # every panel is backed by rows up to the next multiple of 128
        self.plane = self.npanel * self.panel
        self.buf = torch.empty(prec * self.plane, dtype=torch.int16, device=like.device)
        if R % 32:
            # read as k (mode 1) the matrix is walked in whole 32-row K-tiles: the rows that pad the last one must multiply as zeros
            self.buf.view(prec, self.npanel, self.panel // 32, 32)[:, :, R:-(-R // 32) * 32].zero_()
        self.scale_stride = 1 if per_row else 0
        self.inv = torch.empty(R if per_row else 1, dtype=torch.float32, device=like.device) if prec <= 2 else None


def amax_slots(like, want=True):
    """256 zeroed words for a producer kernel's |result|max side output (ud_absmax_commit; the scale of the tensor's planes
    without a pass of its own), or None where the planes path cannot take the tensor"""
    if not want or CFG.spectral_p2 == "off" or like.dtype != torch.float32:
        return None
    return zeros((256,), like)


def split_planes(x2, out=None, prec=3, per_row=False, absmax=None):
    """fp32 [R, C] (row stride >= C) -> Planes: prec 3 the exact three-way bf16 split of gemm_x3.hip, prec 2 two fp16 pieces of
    the scaled matrix — done once by the producer instead of by every workgroup that loads a tile."""
    _chk(x2)
    R, Cc = x2.shape
    assert x2.stride(1) == 1 and Cc % 4 == 0 and x2.stride(0) % 4 == 0
    pl = out if out is not None else Planes(R, Cc, x2, prec, per_row)
    assert pl.R == R and pl.C == Cc
    if pl.prec == 3:
        _call("ud_split_planes", _p(x2), R, Cc, x2.stride(0), _p(pl.buf), pl.panel, pl.plane, _stream())
    elif pl.scale_stride:
        _call("ud_split_planes_h2", _p(x2), R, Cc, x2.stride(0), _p(pl.buf), pl.panel, pl.plane, _p(pl.inv), _stream())
    else:
        amax = absmax                               # 256 partial maxima: the producer's side output, or a pass of our own
        if amax is None:
            amax = empty((256,), x2)
            _call("ud_absmax", _p(x2), R, Cc, x2.stride(0), _p(amax), _stream())
        _call("ud_split_planes_h2t", _p(x2), R, Cc, x2.stride(0), _p(pl.buf), pl.panel, pl.plane, _p(amax), _p(pl.inv),
              _stream())
    return pl


class _WeightPlaneBatch:
    """The planes of every weight matrix the planes GEMMs use, made by TWO launches at the start of a forward
    (ud_split_planes_h2t_multi) instead of an absmax + a split launch per matrix (~50 matrices per step).  A weight registers on
    its first use (an eager step: never inside a graph capture); its planes live in a persistent buffer; begin() re-splits all
    registered weights from their CURRENT values and records each parameter's version, lookup() hands the planes out only while
    that version still holds (an optimizer step in between -> the caller splits the matrix itself) AND only inside the forward
    that made them (begin_forward ... end_forward: the HIP optimizer writes weights through raw pointers, which no version counter
    sees).  One batch per model (kept on the module): a step captured into a hipGraph touches its own model's weights and plane
    buffers only, which live as long as the model; device tables replaced by a rebuild are kept alive for graphs captured with
    them.  Entries die with their parameter (weak references).  cfg.weight_plane_batch = False: off."""

    def __init__(self):
        self.entries = {}          # id(param) -> [weakref(param), Planes, version at the last begin(), shape2, data_ptr]
        self.table = self.slots = None
        self.dirty = False
        self.totals = (0, 0)
        self.active = False        # inside the forward whose begin() made the planes
        self.retired = []          # tables / slots of earlier builds (captured graphs may still launch with them)

    def _base(self, w2):
        b = w2._base if w2._base is not None else w2
        return b if isinstance(b, torch.nn.Parameter) and b.is_contiguous() and b.dtype == torch.float32 else None

    def lookup(self, w2):
        b = self._base(w2)
        if b is None:
            return None
        e = self.entries.get(id(b))
        if e is None or e[0]() is not b or not self.active:
            return None
        return e[1] if (e[2] == b._version and e[3] == tuple(w2.shape) and e[4] == b.data_ptr()) else None

    def register(self, w2):
        b = self._base(w2)
        if b is None or not CFG.weight_plane_batch or torch.cuda.is_current_stream_capturing() or w2.numel() != b.numel():
            return
        e = self.entries.get(id(b))
        if e is not None and e[0]() is b and e[3] == tuple(w2.shape) and e[4] == b.data_ptr():
            return
        import weakref
        R, Cc = w2.shape
        self.entries[id(b)] = [weakref.ref(b), Planes(R, Cc, w2, 2, False), -1, (R, Cc), b.data_ptr()]
        self.dirty = True

    def begin(self):
        self.active = False
        if not self.entries or not CFG.weight_plane_batch:
            return
        # a parameter that died, or whose storage was replaced (p.data = ..., a device move), invalidates the device table
        stale = [k for k, e in self.entries.items() if e[0]() is None or e[0]().data_ptr() != e[4]]
        if stale or self.dirty:
            if torch.cuda.is_current_stream_capturing():
                return                                   # the table is rebuilt on the next eager step; this one splits per matrix
            for k in stale:
                del self.entries[k]
            self.dirty = True
            if not self.entries:
                self.table = self.slots = None
                return
            self._build()
        _call("ud_split_planes_h2t_multi", _p(self.table), len(self.entries), _p(self.slots), self.totals[0], self.totals[1],
              _stream())
        for e in self.entries.values():
            e[2] = e[0]()._version
        self.active = True

    def _build(self):
        from .lib import SplitItem
        items = (SplitItem * len(self.entries))()
        a0 = s0 = 0
        dev = None
        for i, e in enumerate(self.entries.values()):
            b, pl = e[0](), e[1]
            R, Cc = e[3]
            dev = b.device
            it = items[i]
            it.x, it.out, it.inv_scale = b.data_ptr(), pl.buf.data_ptr(), pl.inv.data_ptr()
            it.R, it.ld, it.panel, it.plane, it.C = R, Cc, pl.panel, pl.plane, Cc
            it.amax_block0, it.amax_blocks = a0, max(1, min(256, -(-(R * (Cc // 4)) // 4096)))
            it.split_block0, it.split_bx = s0, -(-R // 64)
            a0 += it.amax_blocks
            s0 += it.split_bx * pl.npanel
            e[2] = -1
        raw = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8)
        if self.table is not None:
            self.retired.append((self.table, self.slots))
        self.table = raw.to(dev)
        self.slots = torch.zeros(256 * len(self.entries), dtype=torch.int32, device=dev)
        self.totals = (a0, s0)
        self.dirty = False


_WEIGHT_PLANES = _WeightPlaneBatch()          # the batch of the forward in progress (default: a process-wide one for direct callers)


class _WeightLayoutBatch:
    """The GEMM layouts of every k x k conv weight of a model ([rows][tap][reduced channel] matrices: forward, and the data
    gradient's flipped / transposed form), made by ONE launch at the start of a forward (ud_weight_layouts_multi) instead of a
    permute / flip + contiguous pair per conv and pass (~45 launches per UDEB4 step, more in the ResNet models).  Same life cycle
    as _WeightPlaneBatch: a (weight, mode) registers on its first eager use; begin() re-makes all registered layouts from the
    CURRENT weights into persistent buffers; get() hands a buffer out only inside that forward and only while the parameter's
    version / storage are the ones it was made from — the backward's layouts are fetched DURING the forward and kept by the
    tape's closures (the weights do not change between a forward and its backward).  One batch per model."""

    def __init__(self):
        self.entries = {}          # (id(param), mode) -> [weakref(param), buffer, version at the last begin(), data_ptr, shape]
        self.table = None
        self.dirty = False
        self.blocks = 0
        self.active = False
        self.retired = []

    def get(self, w, mode):
        if not (isinstance(w, torch.nn.Parameter) and w.is_contiguous() and w.dtype == torch.float32 and w.dim() == 4):
            return None
        e = self.entries.get((id(w), mode))
        if e is not None and e[0]() is w and self.active and e[2] == w._version and e[3] == w.data_ptr():
            return e[1]
        if (e is None or e[0]() is not w or e[3] != w.data_ptr()) and not torch.cuda.is_current_stream_capturing():
            import weakref
            A, B, KH, KW = w.shape
            rows, inner = (A, B) if mode == 0 else (B, A)
            self.entries[(id(w), mode)] = [weakref.ref(w), torch.empty((rows, KH * KW * inner), dtype=torch.float32, device=w.device),
                                           -1, w.data_ptr(), tuple(w.shape)]
            self.dirty = True
        return None

    def begin(self):
        self.active = False
        if not self.entries:
            return
        stale = [k for k, e in self.entries.items() if e[0]() is None or e[0]().data_ptr() != e[3]]
        if stale or self.dirty:
            if torch.cuda.is_current_stream_capturing():
                return
            for k in stale:
                del self.entries[k]
            if not self.entries:
                self.table = None
                return
            self._build()
        _call("ud_weight_layouts_multi", _p(self.table), len(self.entries), self.blocks, _stream())
        for e in self.entries.values():
            e[2] = e[0]()._version
        self.active = True

    def _build(self):
        from .lib import LayoutItem
        items = (LayoutItem * len(self.entries))()
        b0 = 0
        dev = None
        for i, ((_, mode), e) in enumerate(self.entries.items()):
            w = e[0]()
            dev = w.device
            it = items[i]
            it.src, it.dst = w.data_ptr(), e[1].data_ptr()
            it.A, it.B, it.KH, it.KW = e[4]
            it.mode, it.block0 = mode, b0
            b0 += -(-w.numel() // 256)
            e[2] = -1
        if self.table is not None:
            self.retired.append(self.table)
        self.table = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(dev)
        self.blocks = b0
        self.dirty = False


_WEIGHT_LAYOUTS = _WeightLayoutBatch()
_WEIGHT_LAYOUT_BATCH = True          # A/B: tools/run_with.py kernels._WEIGHT_LAYOUT_BATCH=False


def weight_layout(w, mode):
    """W[A][B][KH][KW] as the matrix an implicit-GEMM conv reads — mode 0: [A, KH*KW*B] = permute(0,2,3,1); mode 1: [B, KH*KW*A]
    flipped = flip(2,3).permute(1,2,3,0); mode 2: [B, KH*KW*A] = permute(1,2,3,0) — from the step's one-launch batch when it
    covers this weight as it is now, else made here (and the weight joins the batch from the next forward on)."""
    A, B, KH, KW = w.shape
    if _WEIGHT_LAYOUT_BATCH:
        buf = _WEIGHT_LAYOUTS.get(w, mode)
        if buf is not None:
            return buf
    if mode == 0:
        return w.permute(0, 2, 3, 1).reshape(A, KH * KW * B).contiguous()
    if mode == 1:
        return w.flip(2, 3).permute(1, 2, 3, 0).reshape(B, KH * KW * A).contiguous()
    return w.permute(1, 2, 3, 0).reshape(B, KH * KW * A).contiguous()


def begin_forward(owner=None):
    """start of a model forward: fresh zero blocks, and the planes of all of `owner`'s registered weight matrices in two
    launches (owner: the nn.Module whose forward this is; its batch lives on it)"""
    global _WEIGHT_PLANES, _WEIGHT_LAYOUTS
    reset_zero_pool()
    if owner is not None:
        batch = owner.__dict__.get("_ud_weight_planes")
        if batch is None:
            batch = owner.__dict__["_ud_weight_planes"] = _WeightPlaneBatch()
        _WEIGHT_PLANES = batch
        lay = owner.__dict__.get("_ud_weight_layouts")
        if lay is None:
            lay = owner.__dict__["_ud_weight_layouts"] = _WeightLayoutBatch()
        _WEIGHT_LAYOUTS = lay
    _WEIGHT_PLANES.begin()
    if _WEIGHT_LAYOUT_BATCH:
        _WEIGHT_LAYOUTS.begin()


def end_forward():
    """end of the forward: the batch's planes are not handed out any more (weights may change before the next forward)"""
    _WEIGHT_PLANES.active = False
    _WEIGHT_LAYOUTS.active = False


def weight_batch_snapshot():
    """what the persistent plane / layout buffers of the forward in progress were made from: (batch, {key: parameter version}).
    The backward closures keep raw pointers into those buffers, and every begin_forward re-fills them IN PLACE."""
    return [(b, {k: e[2] for k, e in b.entries.items()}) for b in (_WEIGHT_PLANES, _WEIGHT_LAYOUTS)]


def weight_batch_check(snapshot):
    """raise if a later forward re-filled the buffers from OTHER weight values (forward A, optimizer step, forward B, backward A
    would otherwise compute A's data gradients with B's weights, silently — torch's version counters raise in that sequence)"""
    for b, versions in snapshot:
        for k, v in versions.items():
            e = b.entries.get(k)
            if e is not None and v >= 0 and e[2] >= 0 and e[2] != v:
                raise RuntimeError("backward of a forward whose weight planes / layouts have been re-made from updated weights "
                                   "by a later forward (forward A, optimizer step, forward B, backward A): run A's backward first")


def weight_planes(w2):
    """prec-2 planes (one scale for the tensor) of a conv weight [Cout, Cin]: the step's batch if it covers this weight as it is
    now, else a split of its own (and the weight joins the batch from the next forward on)"""
    pl = _WEIGHT_PLANES.lookup(w2)
    if pl is not None:
        return pl
    _WEIGHT_PLANES.register(w2)
    return split_planes(w2, prec=2)


def p3_ok(M, N, K):
    """shapes ud_gemm_p3 takes and is worth taking: whole 32-deep K-tiles, the large spectral GEMMs"""
    return K % 32 == 0 and min(M, N) >= 128 and K >= 128


# XCD-aware tile raster of ud_gemm_p3's plain / split-K launches: 0x200 | GM << 12 (groups of GM tile rows; see gemm_p3.hip).
# Per shape 0-15 % faster than the round-robin deal (tools/probe_p3_raster.py; GM 2..8 alike), never slower; 0: off
# (A/B: tools/run_with.py unidefense_amd.kernels._P3_RASTER=0 -- python bench.py)
_P3_RASTER = 0x4200
_P3_STAT_SLOT_TILES = 64          # row tiles beyond which the epilogue statistics go through 64 slots + a fold launch


def _p3_desc(A, B, Cout, M, N, K, a_mode, b_mode, out_mode=0, split_k=1, a_row0=0, cfg=0):
    """the ud_gemm_p3_desc of Cout[M][N] (+)= A . B from pre-split operands (see _gemm_p3)"""
    d = GemmP3Desc()
    if a_mode == 0:
        d.A = A.buf.data_ptr() + 2 * (a_row0 * 32)
        d.a_npanel = A.npanel
    else:
        d.A = A.buf.data_ptr() + 2 * ((a_row0 // 32) * A.panel)
        d.a_npanel = A.npanel - a_row0 // 32
    d.B = B.buf.data_ptr()
    d.b_npanel = B.npanel
    d.C = Cout.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.a_panel, d.a_plane, d.b_panel, d.b_plane = A.panel, A.plane, B.panel, B.plane
    d.ldc = N
    d.a_mode, d.b_mode, d.out_mode, d.split_k = a_mode, b_mode, out_mode, split_k
    d.tile_cfg = cfg | (0x100 if _XCD_CONTIGUOUS else 0) | (_P3_RASTER if not cfg & 0x800 else 0)
    assert A.prec == B.prec or max(A.prec, B.prec) <= 2
    d.prec = min(A.prec, B.prec)          # (a prec-1 operand with prec-2 weight planes: the weights' first plane is read)
    d.c_half = 1 if Cout.dtype == torch.float16 else 0
    if d.prec <= 2:          # (prec 1: the first plane of prec-2 planes, one product — the mixed-precision mode)
        assert (a_mode == 0 or not A.scale_stride) and (b_mode == 0 or not B.scale_stride)
        d.a_inv_scale = A.inv.data_ptr() + 4 * a_row0 * A.scale_stride
        d.b_inv_scale = B.inv.data_ptr()
        d.a_scale_stride, d.b_scale_stride = A.scale_stride, B.scale_stride
    return d


_P3_PAIR = True          # A/B: tools/run_with.py kernels._P3_PAIR=False
_P3_PAIR_SK = True       # ... also the shapes whose data gradient alone runs stream-K (as plain tiles in the pair: 25.67 -> 25.39 ms)


_P3_PAIR_ORDER = 2          # 0: the data gradient's workgroups first, 1: the weight gradient's, 2: by rule (A/B below)


def _p3_pair_tn_first(M, N, Kd, pn, pt):
    """which problem's workgroups lead the pair's grid?  A weight gradient with FEW, LONG tiles (reduction over the M pixels per
    split-K slice well above the data gradient's reduction N, at most one round of workgroups) is the pair's critical path: started
    first, the data gradient's many short tiles fill in around it.  Bench step: data gradient first 25.49 ms, weight gradient
    first everywhere 25.51, this rule 25.34 (thresholds x 0.5 ... x 2 within noise; profiles/r05/p3_pair_ab.txt)."""
    if _P3_PAIR_ORDER != 2:
        return _P3_PAIR_ORDER == 1
    sp = int(pt[1]) if pt[0] == "split" else 1
    t1 = -(-N // 128) * -(-Kd // 128) * sp
    return M / sp > 2.0 * N and t1 <= 256


def _p3_pair_ok(pn, pt):
    """can the data gradient and the weight gradient of a 1x1 conv go out as ONE launch of the planes kernel (ud_gemm_p3_pair)?
    Both plans plain or split-K, or a stream-K data gradient (run as plain tiles: the weight gradient's workgroups even out its
    last round instead); no tail form.  Measured on the bench: pairing only where the two grids together fill
    fewer rounds of 256 workgroups than apart (540 + 225 tiles: 3 instead of 3 + 1) 25.85 -> 25.69 ms, pairing always 25.61 — the
    second problem's workgroups start wherever the first's last round leaves a CU free, and a launch is saved.  The weight
    gradient keeps the split-K factor tuned for its own launch: x 0.5 / 0.25 and x 1.5 / 2 / 3 all measured slower (25.4 ->
    25.9 / 26.5 and 25.9 / 26.1 / 26.4 ms, profiles/r05/p3_pair_ab.txt)"""
    return pn[0] in (("plain", "split", "sk") if _P3_PAIR_SK else ("plain", "split")) and pt[0] in ("plain", "split")


def spectral_bwd(ctx, dy2, out=None, dy_absmax=None):
    """(dx, dw) of a 1x1 conv: spectral_dgrad + spectral_wgrad — as ONE launch of the planes kernel where both plans allow
    (ud_gemm_p3_pair: the weight gradient's workgroups follow the data gradient's in one grid)"""
    if (ctx.plans is None or not _P3_PAIR or CFG.deterministic or GEMM_PROFILE is not None or
            not _p3_pair_ok(ctx.plans["nn"], ctx.plans["tn"])):
        dw = spectral_wgrad(ctx, dy2, dy_absmax)
        return spectral_dgrad(ctx, dy2, out=out, dy_absmax=dy_absmax), dw
    dy = _spectral_dy(ctx, dy2, dy_absmax)
    pn, pt = ctx.plans["nn"], ctx.plans["tn"]
    if pn[0] == "sk":
        pn = ("plain",)          # stream-K evens out ONE launch's last round; in the pair the other problem's tiles do
    M, N, Kd = ctx.M, ctx.N, ctx.K
    acc = out is not None
    # nn: dx[M, Kd] = dy[M, N] . w[N, Kd] (reduction N);  tn: dw[N, Kd] = dy[M, N]^T . x[M, Kd] (reduction M)
    if pn[0] == "plain":
        dx = out if acc else empty((M, Kd), dy2, torch.float16 if dy2.dtype == torch.float16 else torch.float32)
        d0 = _p3_desc(dy, ctx.w, dx, M, Kd, -(-N // 32) * 32, 0, 1, 1 if acc else 0, 1)
    else:
        dx = out if acc else split_out((M, Kd), dy2)
        d0 = _p3_desc(dy, ctx.w, dx, M, Kd, -(-N // 32) * 32, 0, 1, 2, int(pn[1]))
    if pt[0] == "plain":
        dw = empty((N, Kd), dy2)
        d1 = _p3_desc(dy, ctx.x, dw, N, Kd, -(-M // 32) * 32, 1, 1, 0, 1)
    else:
        dw = split_out((N, Kd), dy2)
        d1 = _p3_desc(dy, ctx.x, dw, N, Kd, -(-M // 32) * 32, 1, 1, 2, int(pt[1]))
    if _p3_pair_tn_first(M, N, Kd, pn, pt):
        d1.tile_cfg |= 0x10000
    _call("ud_gemm_p3_pair", C.byref(d0), C.byref(d1), _stream())
    return dx, dw


def _gemm_p3(A, B, Cout, M, N, K, a_mode, b_mode, out_mode=0, split_k=1, stats=None, a_row0=0, cfg=0):
    """Cout[M][N] (+)= A . B from pre-split operands.  a_row0: first GEMM row of A used (a multiple of 128; mode 0: a row
    offset inside every panel, mode 1: whole panels).  stats as in _gemm."""
    d = _p3_desc(A, B, Cout, M, N, K, a_mode, b_mode, out_mode, split_k, a_row0, cfg)
    slices = None
    if CFG.deterministic and out_mode == 2:
        total = M * N
        fresh = getattr(Cout, "_ud_fresh", False)
        ws = _slice_ws(Cout, split_k * total)
        d.C, d.out_mode, d.slice_stride = ws.data_ptr(), 3, total
        slices = (ws, total, total, 0 if fresh else 1)
        if fresh:
            Cout._ud_fresh = False
    fold = None
    stats_done = False
    if stats is not None and _GEMM_EPILOGUE_STATS and out_mode == 0 and split_k == 1:
        stats_done = True
        slots = 64 if -(-M // 128) > _P3_STAT_SLOT_TILES else 1          # (gemm_p3.hip's epilogue applies the same rule)
        tgt = stats if slots == 1 else zeros64(2 * slots * N, Cout)
        d.stat_sum, d.stat_sumsq = tgt.data_ptr(), tgt.data_ptr() + 8 * (slots * N if slots > 1 else N)
        if slots > 1:
            fold = (tgt, slots)
    if GEMM_PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _call("ud_gemm_p3", C.byref(d), _stream())
        e1.record()
        GEMM_PROFILE.append((e0, e1, 2.0 * M * N * K, (M, N, K, a_mode, b_mode, split_k, 1), 4,
                             2.0 * A.prec * (M * K + K * N) + 4.0 * M * N))
    else:
        _call("ud_gemm_p3", C.byref(d), _stream())
    if slices is not None:
        ws, stride, total, accumulate = slices
        _call("ud_sum_slices", _p(ws), _p(Cout), split_k, total, stride, accumulate, _stream())
    if fold is not None:
        tgt, slots = fold
        _call("ud_stat_slots_fold", _pd64(tgt), _pd64(tgt, slots * N), slots, N, _pd64(stats), _pd64(stats, N), _stream())
    return (Cout, stats_done) if stats is not None else Cout


_TAIL_SPLIT = True
_XCD_CONTIGUOUS = False          # ud_gemm_desc.tile_cfg bit 8: each XCD takes a contiguous range of the tile order
_GEMM_EPILOGUE_STATS = True


def _pd64(t, off_doubles=0):
    return C.c_void_p(t.data_ptr() + 8 * off_doubles)


def _tail_plan(M, N, K):
    """Tile quantisation: 540 tiles of 128x128 on 256 CUs take 3 rounds of which the last is 11 % full (the
    4608x1920x1920 spectral convs: 131 instead of ~180 TFLOP/s).  Plan: the leading row-tiles that fill whole
    rounds run as one plain launch, the few remaining row-tiles as a second, split-K launch whose blocks are 1/s as
    long.  Returns (rows of the plain part, split) or None.  (Assumes the 128x128 configuration gemm_x3.hip picks
    for shapes this large; a different pick only costs the gain.)"""
    if not _TAIL_SPLIT or M < 1024 or N < 128 or K < 512:
        return None
    mt, nt = -(-M // 128), -(-N // 128)
    full, tail = divmod(mt * nt, 256)
    if full < 1 or tail == 0:
        return None
    rows_tail = -(-tail // nt)
    tail_tiles = rows_tail * nt
    if tail_tiles > 64 or rows_tail >= mt:
        return None
    split = min(8, 256 // tail_tiles, K // 256)
    if split < 2:
        return None
    return (mt - rows_tail) * 128, split


# ---- per-shape launch tuner of the plain fp32 GEMMs ---------------------------------------------------------------
# The model's ~90 GEMM shapes are fixed; the cost models above (tile rounds, split-K targets) miss the best (tile, split-K)
# of many of them — the thin expand / project shapes are bound by the latency of a workgroup's k-loop, where 64x64
# tiles with a few splits win, and e.g. 1152x3264x3264 runs 11 % faster split in two.  So the FIRST eager call with a
# shape measures the candidates (a handful of back-to-back launches each, into scratch) and the winner is cached; an
# exhaustive offline sweep (tools/sweep_gemm_plans.py) put the gain at 1.2 ms of the 17.8 ms the plain GEMMs take per
# step.  Never inside a graph capture (an unseen shape then takes the cost-model plan); UD_GEMM_TUNE=0 turns it off.
_TUNED = {}
# UD_GEMM_TUNE_CACHE=<file>: plans are read from / added to this JSON file, so that a profiled run (rocprofv3, PMC passes)
# repeats the plans of the benchmarked one without the tuner's measurement launches in its kernel statistics
# Shipped defaults: the plans measured on an MI355X for the shapes of the BASELINE configs (bs 32 / 64 UDEB4, UDR18, UDR50,
# the engine's train step) and of the parity tests — those shapes start with a plan instead of a measurement (the same
# plan in every run: repeatable rounding); anything else is tuned on first use.
# UD_GEMM_TUNE_DEFAULTS=0 ignores the file (every shape is measured on this machine).
_TUNE_DEFAULTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_plans_gfx950.json")


def _load_plans(path):
    import json
    with open(path) as fh:
        return {tuple(json.loads(k)): (tuple(v) if v is not None else None) for k, v in json.load(fh).items()}


if CFG.gemm_tune_defaults and os.path.exists(_TUNE_DEFAULTS):
    _TUNED.update(_load_plans(_TUNE_DEFAULTS))
if CFG.gemm_tune_cache and os.path.exists(CFG.gemm_tune_cache):
    _TUNED.update(_load_plans(CFG.gemm_tune_cache))


def _tune_cache_save():
    path = CFG.gemm_tune_cache
    if not path:
        return
    import json
    tmp = path + ".tmp%d" % os.getpid()
    with open(tmp, "w") as fh:
        json.dump({json.dumps(list(k)): (list(v) if v is not None else None) for k, v in _TUNED.items()}, fh)
    os.replace(tmp, path)
_X3_TILES = {1: (128, 128), 2: (128, 64), 3: (64, 128), 4: (64, 64)}
_TUNE_SPLITS = (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 128, 256, 384)


def _tune_candidates(M, N, K):
    heavy = 2.0 * M * N * K > 2.0e10                   # the large spectral GEMMs: a few plans only
    out = []
    for cfg, (bm, bn) in _X3_TILES.items():
        tiles = -(-M // bm) * -(-N // bn)
        for split in _TUNE_SPLITS:
            if split > 1 and (K // split < 64 or tiles * split > 4096 or (heavy and split > 4)):
                continue
            if (tiles * split < 64 and K >= 1024) or (heavy and cfg == 4):
                continue
            out.append((cfg, split))
    return out


_TUNE_N = 6          # launches per timing graph


def _time_launches(fn, n=None):
    """Device time of one fn() in ms.  The launches are replayed from a small hipGraph: eager launches of a 20 us kernel
    are paced by the host (ctypes + Python), which hides the differences the tuner is after."""
    n = _TUNE_N if n is None else n
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cur = torch.cuda.current_stream()
    try:
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
                for _ in range(n):
                    fn()
        torch.cuda.synchronize()
        with torch.cuda.stream(s):
            g.replay()
            e0.record()
            g.replay()
            g.replay()
            e1.record()
        e1.synchronize()
        cur.wait_stream(s)
        return e0.elapsed_time(e1) / (2 * n)
    except Exception:                                   # noqa: BLE001 — no capture possible here: eager timing
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / n


def _tuned_plan(kind, M, N, K, launch, baseline, extra_if_split=None, no_split=False):
    """(cfg, split) to run this shape with, or None = keep the cost-model plan.  launch(cfg, split): enqueue one launch
    into scratch; baseline(): enqueue the cost-model plan; extra_if_split(): work a split plan adds (ud_colstats when the
    caller wanted epilogue statistics); no_split: a half result takes no atomics — tiles only."""
    kind = kind if isinstance(kind, str) else "/".join(str(v) for v in kind)
    key = (kind, M, N, K, extra_if_split is not None, _call("ud_gemm_get_path"))
    if key in _TUNED:
        return _TUNED[key]
    if not CFG.gemm_tune or torch.cuda.is_current_stream_capturing() or min(M, N, K) < _X3_MINDIM:
        return None
    best_t, best = _time_launches(baseline), None
    extra = _time_launches(extra_if_split) if extra_if_split is not None else 0.0
    for cfg, split in _tune_candidates(M, N, K):
        if no_split and split > 1:
            continue
        t = _time_launches(lambda: launch(cfg, split)) + (extra if split > 1 else 0.0)
        if t < 0.97 * best_t:                          # a clear win only: equal plans keep the model's choice
            best_t, best = t, (cfg, split)
    _TUNED[key] = best
    _tune_cache_save()
    return best


def _model_plan_launch(a, w, out, M, N, K, lda, ldb, b_mode, acc):
    """The cost-model plan of a forward / data-gradient GEMM: tail split, split-K for launches that cannot fill the
    chip, or one plain launch.  out: existing gradient to accumulate onto (acc) or None (a fresh result is returned)."""
    plan = _tail_plan(M, N, K)
    if plan is not None:
        m1, split = plan
        if not acc:
            out = empty((M, N), a)
        _gemm(a, w, out, m1, N, K, lda, ldb, N, 0, b_mode, 1 if acc else 0)
        tail = out[m1:]
        if not acc:
            tail.zero_()
        _gemm(a[m1:], w, tail, M - m1, N, K, lda, ldb, N, 0, b_mode, 2, split)
        return out
    split = _fwd_split(M, N, K)
    if split > 1:
        if not acc:
            out = split_out((M, N), a)
        return _gemm(a, w, out, M, N, K, lda, ldb, N, 0, b_mode, 2, split)
    if not acc:
        out = empty((M, N), a)
    return _gemm(a, w, out, M, N, K, lda, ldb, N, 0, b_mode, 1 if acc else 0)


def _tuned_launch(kind, a, w, out, M, N, K, lda, ldb, a_mode, b_mode, acc, stats, model, out_dtype=torch.float32):
    """Run the measured-best (tile, split-K) of this shape if the tuner has / can find one, else model().
    out_dtype: torch.float16 for the half-storage forward / data-gradient products (plain launches only)."""
    tmp = []
    kind = kind if out_dtype == torch.float32 and a.dtype == torch.float32 else kind + "/h"

    def scratch():
        if not tmp:
            tmp.append(empty((M, N), a, out_dtype))
        return tmp[0]
    sacc = []

    def stats_pass():                                   # what a split plan adds when the caller wants epilogue statistics
        if not sacc:
            sacc.append(torch.zeros(2 * N, dtype=torch.float64, device=a.device))
        colstats(scratch(), sacc[0])
    tuned = _tuned_plan(kind, M, N, K,
                        lambda cfg, split: _gemm(a, w, scratch(), M, N, K, lda, ldb, N, a_mode, b_mode,
                                                 2 if split > 1 else 0, split, cfg=cfg),
                        lambda: model(scratch()),
                        stats_pass if stats is not None else None, no_split=out_dtype != torch.float32)
    if tuned is None:
        return None
    cfg, split = tuned
    if split == 1:
        if not acc:
            out = empty((M, N), a, out_dtype)
        return _gemm(a, w, out, M, N, K, lda, ldb, N, a_mode, b_mode, 1 if acc else 0, stats=stats, cfg=cfg)
    if not acc:
        out = split_out((M, N), a)
    r = _gemm(a, w, out, M, N, K, lda, ldb, N, a_mode, b_mode, 2, split, cfg=cfg)
    return (r, False) if stats is not None else r


def gemm_nt(a, w, out=None, accumulate=False, stats=None):
    """out[M,N] (+)= a[M,K] @ w[N,K]^T     (1x1 conv / linear forward).
    stats (2N zeroed doubles): BatchNorm statistics of the result; returns (out, done) — done = the GEMM epilogue
    accumulated them (plain launches only: a split-K or tail-split plan leaves them to ud_colstats)."""
    half = _act(a)
    _chk(w)
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K
    if half:
        # half storage: one plain launch (no atomics onto a half result), its tile tuned like the fp32 shapes'
        if out is None:
            r = _tuned_launch("nt", a, w, None, M, N, K, K, K, 0, 0, False, stats,
                              lambda scr: _gemm(a, w, scr, M, N, K, K, K, N, 0, 0, 0), out_dtype=a.dtype)
            if r is not None:
                return r
            out = empty((M, N), a, a.dtype)
        return _gemm(a, w, out, M, N, K, K, K, N, 0, 0, 1 if accumulate else 0, stats=stats)
    if out is None:
        # during tuning the model plan is timed into scratch: as a plain launch there, its splits need no zeroing
        r = _tuned_launch("nt", a, w, None, M, N, K, K, K, 0, 0, False, stats,
                          lambda scr: _model_plan_launch(a, w, scr, M, N, K, K, K, 0, True))
        if r is not None:
            return r
    if stats is not None:
        assert out is None and not accumulate
        if _tail_plan(M, N, K) is None and _fwd_split(M, N, K) <= 1:
            return _gemm(a, w, empty((M, N), a), M, N, K, K, K, N, 0, 0, 0, stats=stats)
        return _model_plan_launch(a, w, None, M, N, K, K, K, 0, False), False
    if out is None:
        return _model_plan_launch(a, w, None, M, N, K, K, K, 0, False)
    return _gemm(a, w, out, M, N, K, K, K, N, 0, 0, 1 if accumulate else 0)


def gemm_nn(a, w, out=None, accumulate=False):
    """out[M,N] (+)= a[M,K] @ w[K,N]       (data gradient of a 1x1 conv: dY @ W)"""
    half = _act(a, out)
    _chk(w)
    M, K = a.shape
    N = w.shape[1]
    assert w.shape[0] == K
    if half:
        acc = out is not None and accumulate
        if out is None or acc:
            r = _tuned_launch("nn", a, w, out, M, N, K, K, N, 0, 1, acc, None,
                              lambda scr: _gemm(a, w, scr, M, N, K, K, N, N, 0, 1, 0), out_dtype=a.dtype)
            if r is not None:
                return r
        if out is None:
            out = empty((M, N), a, a.dtype)
        return _gemm(a, w, out, M, N, K, K, N, N, 0, 1, 1 if acc else 0)
    if out is None or accumulate:
        # accumulate: `out` already holds a term of the same gradient (the skip branch's): the plain part adds into
        # it (out_mode 1), the split-K parts add atomically onto it — no zero fill, no separate axpby pass
        acc = out is not None
        r = _tuned_launch("nn", a, w, out, M, N, K, K, N, 0, 1, acc, None,
                          lambda scr: _model_plan_launch(a, w, scr, M, N, K, K, N, 1, True))
        if r is not None:
            return r
        return _model_plan_launch(a, w, out, M, N, K, K, N, 1, acc)
    return _gemm(a, w, out, M, N, K, K, N, N, 0, 1, 0)


# ---- spectral 1x1 convs on pre-split fp16 x 2 planes (ud_gemm_p3 prec 2) -------------------------------------------------
# F.conv2d(x_freq, freq_conv.weight) of the SF blocks (model/efficientnet/exp.py:57), its data gradient and its weight gradient:
# three products over the SAME three matrices (x_freq [pixels x 2C], the weight [2C x 2C], dY [pixels x 2C]).  Each matrix is
# split ONCE into two fp16 pieces with one power-of-two scale for the tensor (ud_absmax + ud_split_planes_h2t; the planes then
# serve every GEMM mode) and the products run on three fp16 MFMAs per tile instead of six bf16 ones; the fp32 tensors are not
# kept.  Which blocks take this path, and each product's launch plan, is measured per shape on first use (`p2c` entries of the
# plan table) like the tile / split-K plans of the in-kernel-split GEMMs.
_P2_MIN = (1024, 512)          # untuned `auto`: M and min(N, K) from which the planes path is taken
_P2_SPLITS = (2, 3, 4, 6, 8)
_P2_TN_SPLITS = (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256)
_P2_MODES = {"nt": (0, 0), "nn": (0, 1), "tn": (1, 1)}


def _p2_shape_ok(M, N, K):
    """a 1x1 conv y[M,N] = x[M,K] w[N,K]^T on the planes kernel: its three products reduce over K, N and M in 32-deep K-tiles —
    channel counts are padded with zero columns / zero weight rows by the split, pixel counts must be whole tiles"""
    return M % 32 == 0 and K % 4 == 0 and N % 4 == 0 and min(N, K) >= 64 and M >= 128


def _p2_plans(kind, M, N, K):
    tiles = -(-M // 128) * -(-N // 128)
    out = []
    if kind == "tn":
        for s_ in _P2_TN_SPLITS:
            if (s_ == 1 or K // 32 // s_ >= 4) and 64 <= tiles * s_ <= 2048:
                out.append(("split", s_) if s_ > 1 else ("plain",))
    else:
        out.append(("plain",))
        for s_ in _P2_SPLITS:
            if K // 32 // s_ >= 4 and tiles * s_ <= 2048:
                out.append(("split", s_))
        tp = _tail_plan(M, N, K)
        if tp is not None:
            out.append(("tail", tp[0], tp[1]))
        if kind == "nt" and not CFG.deterministic:
            out += [("tailp", m1, s_) for m1, s_ in _tailp_plans(M, N, K)]
    if not CFG.deterministic:
        out.append(("sk",))
    return out or [("plain",)]


_TAILP = True          # A/B: tools/run_with.py kernels._TAILP=False


def _tailp_plans(M, N, K):
    """the tail PAIR of a forward product (ud_gemm_p3_pair, round 6): (rows of the plain part, split-K of the remaining row tiles)
    candidates — the row tiles beyond whole rounds of the 256 CUs go out in the SAME grid as short split-K workgroups"""
    if not _TAILP or M < 256 or N < 128 or K < 256:
        return []
    mt, nt = -(-M // 128), -(-N // 128)
    full, tail = divmod(mt * nt, 256)
    if full < 1 or tail == 0:
        return []
    out = []
    for extra in (0, 1):                                  # one more row tile in the tail: the plain part just UNDER whole rounds
        rows_tail = -(-tail // nt) + extra
        tail_tiles = rows_tail * nt
        if rows_tail >= mt or tail_tiles > 224:
            continue
        for s_ in sorted({max(2, min(16, 256 // tail_tiles)), max(2, min(16, 512 // tail_tiles))}):
            if K // 32 // s_ >= 4:
                out.append(((mt - rows_tail) * 128, s_))
    return out


def _p2_run(kind, plan, ap, bp, M, N, K, like, stats=None, out=None):
    """one product of the planes path.  kind nt: ap [M,K] x bp [N,K]; nn: ap [M,K] x bp [K,N]; tn: ap [K,M] x bp [K,N].
    out: an existing term the product is ADDED to (the skip branch's gradient), else a fresh result.  stats: see _gemm (only a
    plain launch fills them: returns (out, done))."""
    am, bm = _P2_MODES[kind]
    Kp = -(-K // 32) * 32                      # the operands' zero padding makes up the last K-tile
    how = plan[0]
    if how == "tailp" and GEMM_PROFILE is not None:
        how = "tail"          # the instrumented steps time every product through _gemm_p3: the same two problems as two launches
    acc = out is not None
    assert not (acc and stats is not None)
    # the mixed-precision mode (prec-1 operands of half-stored activations): activations come out half-stored too, weight
    # gradients fp32; half results take plain launches only
    odt = torch.float16 if (like.dtype == torch.float16 and kind != "tn") else torch.float32
    assert odt == torch.float32 or how == "plain"
    if how == "plain":
        return _gemm_p3(ap, bp, out if acc else empty((M, N), like, odt), M, N, Kp, am, bm, 1 if acc else 0, 1, stats=stats)
    res = None
    if how == "split":
        res = _gemm_p3(ap, bp, out if acc else split_out((M, N), like), M, N, Kp, am, bm, 2, int(plan[1]))
    elif how == "sk":
        res = _gemm_p3(ap, bp, out if acc else zeros((M, N), like), M, N, Kp, am, bm, 1 if acc else 0, 1, cfg=0x800)
    elif how == "tailp":
        # the tail pair (nt): plain leading row tiles + split-K last row tiles as ONE grid (ud_gemm_p3_pair)
        m1, s_ = int(plan[1]), int(plan[2])
        assert kind == "nt"          # (statistics wanted: left to ud_colstats like every non-plain plan — (res, False) below)
        res = out if acc else empty((M, N), like)
        tail = res[m1:]
        if not acc:
            tail.zero_()
        d0 = _p3_desc(ap, bp, res, m1, N, Kp, am, bm, 1 if acc else 0, 1)
        d1 = _p3_desc(ap, bp, tail, M - m1, N, Kp, am, bm, 2, s_, a_row0=m1)
        _call("ud_gemm_p3_pair", C.byref(d0), C.byref(d1), _stream())
    else:
        m1, s_ = int(plan[1]), int(plan[2])          # "tail" (nt / nn): whole rounds of tiles plain, the last row tiles split
        res = out if acc else empty((M, N), like)
        _gemm_p3(ap, bp, res, m1, N, Kp, am, bm, 1 if acc else 0, 1)
        tail = res[m1:]
        if not acc:
            if CFG.deterministic:
                tail._ud_fresh = True
            else:
                tail.zero_()
        _gemm_p3(ap, bp, tail, M - m1, N, Kp, am, bm, 2, s_, a_row0=m1)
    return (res, False) if stats is not None else res


def _p2_default_plan(kind, M, N, K):
    """untuned: a plain launch unless the tile count sits just above whole rounds of the CUs; weight gradients split"""
    tiles = -(-M // 128) * -(-N // 128)
    if kind == "tn":
        s_ = max(1, min(256 // tiles if tiles <= 256 else 1, K // 32 // 4))
        return ("split", s_) if s_ > 1 else ("plain",)
    full, rem = divmod(tiles, 256)
    if full >= 1 and 0 < rem <= 64:
        tp = _tail_plan(M, N, K)
        if tp is not None:
            return ("tail", tp[0], tp[1])
        if not CFG.deterministic:
            return ("sk",)
    if tiles < 128 and K >= 1024:
        return ("split", min(4, max(2, 256 // tiles)))
    return ("plain",)


_HALF_LIKE = {}


def _half_like(t):
    """an (empty) half tensor on t's device: the `like` of launches whose activation results are half-stored"""
    h = _HALF_LIKE.get(t.device)
    if h is None:
        h = _HALF_LIKE[t.device] = torch.empty(0, dtype=torch.float16, device=t.device)
    return h


def spectral_takes_plane_half(M, N, Kd):
    """does the 1x1 conv [M, Kd] x [N, Kd]^T of the mixed-precision mode run on the planes kernel (prec 1)?"""
    return _p1_plans_for(M, N, Kd) is not None


def rfft2_plane_half_ok(x):
    """can ud_rfft2_ex_plane_half lay this half-stored transform's result into the spectral GEMM's plane?"""
    if not (_P1_PLANES and _P1_DIRECT and _RFFT_PLANES and x.dtype == torch.float16 and CFG.spectral_p2 != "off"):
        return False
    N, S, _, Cc = x.shape
    return S in (8, 16, 32, 12, 24, 48) and (2 * Cc) % 32 == 0 and not _fft_two_pass("rfft_ex", S, 1)


def rfft2_ex_plane_half(x, scale, w_interior=1.0, bn=None, want_act=False, gate_alpha=None, gate_mode=0, update=False,
                        gate_acc=None, dw_wt=None, dw_k=0):
    """rfft2_ex of a half-stored x whose half result goes straight into the ONE plane ud_gemm_p3 prec 1 reads (scale 1).
    dw_wt / dw_k: also the stride-1 depthwise conv of the activated plane (appended to the result).
    Returns (Planes, activated input or None[, gate gradient][, conv result])."""
    _act(x)
    N, S, S2, Cc = x.shape
    assert S == S2 and x.dtype == torch.float16
    pl = Planes(N * S * (S // 2 + 1), 2 * Cc, x, 1, False)
    act = torch.empty_like(x) if (want_act and bn is not None) else None
    ggrad = empty((), x) if gate_acc is not None else None
    spat = torch.empty_like(x) if dw_k else None
    _call("ud_rfft2_ex_plane_half", _p(x), _p(pl.buf), pl.panel, _p(pl.inv), N, S, Cc, float(scale), float(w_interior),
          C.byref(bn.ref(update)) if bn is not None else None, _p(act), _p(gate_alpha), int(gate_mode),
          _pd(gate_acc) if gate_acc is not None else None, _p(ggrad), _p(dw_wt), _p(spat), int(dw_k), _stream())
    out = (pl, act, ggrad) if gate_acc is not None else (pl, act)
    return out + (spat,) if dw_k else out


def planes_from_half(x2):
    """a half-stored [R, C] matrix as the ONE fp16 plane ud_gemm_p3 prec 1 reads (P32 layout, values unchanged)"""
    R, Cc = x2.shape
    assert x2.is_cuda and x2.dtype == torch.float16 and x2.stride(1) == 1 and x2.stride(0) % 8 == 0
    pl = Planes(R, Cc, x2, 1, False)
    _call("ud_planes_from_half", _p(x2), R, Cc, x2.stride(0), _p(pl.buf), pl.panel, _p(pl.inv), _stream())
    return pl


# The mixed-precision mode (ud_gemm path 3, BASELINE configs[4]) on the planes kernel: one fp16 plane per operand, one product per
# tile — 530-770 TFLOP/s on the spectral convs' shapes against 290-530 of gemm_x3_kernel's fp16 form (tools/bench_p3_prec1.py).
# The half-stored activation becomes a plane by a layout pass (ud_planes_from_half), the weights' planes come from the step's
# batch (their first plane), half results are stored by the epilogue: plain launches only (no atomics onto half).
_P1_PLANES = True          # A/B: tools/run_with.py kernels._P1_PLANES=False
_P1_DIRECT = True          # ... the producers lay half results into the plane themselves (no ud_planes_from_half pass)
_P1_DW = True              # ... and the transform also computes the stride-1 depthwise conv (as in the fp32 mode)
_P1_MIN = (1024, 128)      # M, min(N, K) from which the layout pass pays (f16 bs 64: (1024, 512) 32.5 ms, (1024, 256) 32.0,
#                            (1024, 128) 31.9, (4096, 64) 33.2; without the path 35.4 — profiles/r05/f16_p1_planes_ab.txt)


def _p1_plans_for(M, N, Kd):
    if not (_P1_PLANES and CFG.spectral_p2 != "off" and _call("ud_gemm_get_path") == 3 and _p2_shape_ok(M, N, Kd) and
            Kd % 8 == 0 and N % 8 == 0 and M >= _P1_MIN[0] and min(N, Kd) >= _P1_MIN[1]):
        return None
    return {"nt": ("plain",), "nn": ("plain",), "tn": _p2_default_plan("tn", N, Kd, M)}


class SpectralCtx:
    """what the backward of one 1x1 conv needs: the planes (or, on the in-kernel-split path, the fp32 operands)"""
    __slots__ = ("plans", "x", "w", "dy", "M", "N", "K")


def _p2_block_plans(x2, w2, want_stats=False):
    """None (in-kernel split) or {"nt": plan, "nn": plan, "tn": plan} for a 1x1 conv of this shape.  want_stats: the forward
    launch is to fill BatchNorm statistics in its epilogue (a plain launch does; other plans are charged a ud_colstats pass)"""
    M, Kd = x2.shape
    if x2.dtype != torch.float32:
        return _p1_plans_for(M, w2.shape[0], Kd) if (x2.dtype == torch.float16 and w2.dtype == torch.float32) else None
    return _p2_plans_for(M, w2.shape[0], Kd, want_stats, w2, x2)


def _p2_plans_for(M, N, Kd, want_stats, w2, x2, tune=True, force=False):
    """x2: the fp32 operand the on-line tuner may measure with (None: a producer-made Planes operand / a query — an unknown shape
    then takes the untuned rule)"""
    mode = CFG.spectral_p2
    if (mode == "off" or w2.dtype != torch.float32 or not _p2_shape_ok(M, N, Kd) or
            _call("ud_gemm_get_path") not in (0, 2)):
        return None
    key = ("p2c", M, N, Kd, bool(CFG.deterministic), mode == "on", bool(want_stats))
    plans = _TUNED.get(key, "?")
    if plans is None and force:
        plans = "?"          # a producer-made operand (im2col planes) has no other path: a table entry that judged a 1x1 conv of
        tune = False         # the same M, N, K slower on planes does not apply — take the default plans
    if plans == "?":
        if tune and x2 is not None and CFG.gemm_tune and not torch.cuda.is_current_stream_capturing():
            plans = _p2_tune(key, x2, w2, M, N, Kd, mode == "on", want_stats)
        elif mode == "on" or force or (M >= _P2_MIN[0] and min(N, Kd) >= _P2_MIN[1]):
            plans = [("plain",) if want_stats else _p2_default_plan("nt", M, N, Kd), _p2_default_plan("nn", M, Kd, N),
                     _p2_default_plan("tn", N, Kd, M)]
        else:
            plans = None
    if plans is None:
        return None
    return {"nt": tuple(plans[0]), "nn": tuple(plans[1]), "tn": tuple(plans[2])}


def spectral_fwd(x2, w2, stats=None, x_absmax=None, force=False):
    """y[M, N] = x[M, K] @ w[N, K]^T (a 1x1 conv; the spectral convs are the square case) and the context of its backward.
    stats: BatchNorm accumulator of the result (gemm_nt's contract: returns ((y, done), ctx)).
    x2: the fp32 matrix, or Planes its producer wrote itself (only for shapes spectral_takes_planes accepts)."""
    ctx = SpectralCtx()
    ctx.N = w2.shape[0]
    ctx.dy = None
    if isinstance(x2, Planes):
        ctx.M, ctx.K = x2.R, x2.C
        if x2.prec == 1:          # the mixed-precision mode: a half-stored activation its producer laid into the plane
            ctx.plans = _p1_plans_for(ctx.M, ctx.N, ctx.K)
            like = _half_like(w2)
        else:
            ctx.plans = _p2_plans_for(ctx.M, ctx.N, ctx.K, stats is not None, w2, None, force=force)
            like = w2
        assert ctx.plans is not None
        ctx.x, ctx.w = x2, weight_planes(w2)
        return _p2_run("nt", ctx.plans["nt"], ctx.x, ctx.w, ctx.M, ctx.N, ctx.K, like, stats=stats), ctx
    ctx.M, ctx.K = x2.shape
    ctx.plans = _p2_block_plans(x2, w2, stats is not None)
    if ctx.plans is None:
        ctx.x, ctx.w = x2, w2
        return gemm_nt(x2, w2, stats=stats), ctx
    if x2.dtype == torch.float16:
        ctx.x, ctx.w = planes_from_half(x2), weight_planes(w2)
    else:
        ctx.x, ctx.w = split_planes(x2, prec=2, absmax=x_absmax), weight_planes(w2)
    return _p2_run("nt", ctx.plans["nt"], ctx.x, ctx.w, ctx.M, ctx.N, ctx.K, x2, stats=stats), ctx


def spectral_takes_planes(M, N, Kd, w2, want_stats=False):
    """does the 1x1 conv [M, Kd] x [N, Kd]^T run on the planes GEMM with the plans known NOW (shipped / cached / heuristic — no
    tuning launch: a producer asks before it decides how to write its result)?"""
    return _p2_plans_for(M, N, Kd, want_stats, w2, None, tune=False) is not None


def _spectral_dy(ctx, dy2, absmax=None):
    if ctx.dy is None:
        ctx.dy = planes_from_half(dy2) if dy2.dtype == torch.float16 else split_planes(dy2, prec=2, absmax=absmax)
    return ctx.dy


def spectral_dgrad(ctx, dy2, out=None, dy_absmax=None):
    """dx[M, K] = dy[M, N] @ w[N, K]   (out: a term of the same gradient to add onto, in place)"""
    if ctx.plans is None:
        return gemm_nn(dy2, ctx.w, out=out, accumulate=out is not None)
    return _p2_run("nn", ctx.plans["nn"], _spectral_dy(ctx, dy2, dy_absmax), ctx.w, ctx.M, ctx.K, ctx.N, dy2, out=out)


def spectral_wgrad(ctx, dy2, dy_absmax=None):
    """dw[N, K] = dy[M, N]^T @ x[M, K]"""
    if ctx.plans is None:
        return gemm_tn(dy2, ctx.x)
    return _p2_run("tn", ctx.plans["tn"], _spectral_dy(ctx, dy2, dy_absmax), ctx.x, ctx.N, ctx.K, ctx.M, dy2)


def _p2_tune(key, x2, w2, M, N, Kd, forced, want_stats=False):
    """the three products of a 1x1 conv on the in-kernel-split path against the planes path (its three splits included),
    every plan of each product measured; the winner is cached (None = in-kernel split)"""
    dy2 = torch.randn(M, N, device=x2.device)
    acc = torch.zeros(2 * N, dtype=torch.float64, device=x2.device) if want_stats else None

    def fwd_x3():
        r = gemm_nt(x2, w2, stats=acc)
        if want_stats and not r[1]:
            colstats(r[0], acc)
    t_x3 = _time_launches(lambda: (fwd_x3(), gemm_nn(dy2, w2), gemm_tn(dy2, x2)))
    xp, wp, dp = split_planes(x2, prec=2), split_planes(w2, prec=2), split_planes(dy2, prec=2)
    t_p2 = _time_launches(lambda: (split_planes(x2, xp), split_planes(dy2, dp), split_planes(w2, wp)))
    t_stats = _time_launches(lambda: colstats(dy2, acc)) if want_stats else 0.0
    plans = []
    for kind, (a, b, m, n, k) in (("nt", (xp, wp, M, N, Kd)), ("nn", (dp, wp, M, Kd, N)), ("tn", (dp, xp, N, Kd, M))):
        best, best_t = None, 1e30
        for plan in _p2_plans(kind, m, n, k):
            st = acc if (kind == "nt" and want_stats) else None
            t = _time_launches(lambda: _p2_run(kind, plan, a, b, m, n, k, x2, stats=st))
            if st is not None and plan[0] != "plain":
                t += t_stats
            if t < best_t:
                best, best_t = plan, t
        plans.append(list(best))
        t_p2 += best_t
    res = plans if (forced or t_p2 < 0.97 * t_x3) else None
    _TUNED[key] = res
    _tune_cache_save()
    return res


_WG_SPLIT_MAXT = 512
_WG_SPLIT_TARGET = 768
_WG_SPLIT_ROWS = 256


def _pick_split(tiles, K):
    """split-K factor for a weight-gradient GEMM whose reduction runs over K pixels: enough workgroups to
    fill 256 CUs about three times over, at least 256 reduction rows per split."""
    if tiles >= _WG_SPLIT_MAXT or K < 512:
        return 1
    s = min(max(1, K // _WG_SPLIT_ROWS), -(-_WG_SPLIT_TARGET // tiles))
    return max(1, min(s, 1024))


_TILE_CFGS = ((128, 128, 1.00), (128, 64, 0.93), (256, 32, 1.10), (32, 256, 1.10), (64, 128, 0.93))
_X3_TILE_MODEL = False      # A/B: the fp32-kernel tile model (more split-K) is 0.4 % faster


_FWD_SPLIT_T = 224      # A/B on the bench: 128/1024 -> 224/512 = -0.6 % step time
_FWD_SPLIT_K = 512
_FWD_SPLIT_MINK = 256     # reduction length per split, at least
_FWD_SPLIT_WGS = 320       # workgroups the split aims at


def _fwd_split(M, N, K):
    """split-K for a forward / data-gradient GEMM too small to fill the chip (e.g. the 8x8-resolution
    expand/project convs: M = 2048, 80 tiles for 256 CUs)."""
    t = _tiles(M, N, K)
    if t >= _FWD_SPLIT_T or K < _FWD_SPLIT_K:
        return 1
    return max(1, min(K // _FWD_SPLIT_MINK, -(-_FWD_SPLIT_WGS // t)))


_X3_CFGS = ((128, 128, 1.00), (128, 64, 1.20), (64, 128, 1.20))          # gemm_x3.hip kX
_X3_MINDIM = 16                 # gemm.hip's auto rule


def _tiles(M, N, K=None):
    """Tile count of the configuration the library picks for an unsplit launch: gemm_x3.hip's for plain GEMMs whose
    dimensions all reach the auto rule's minimum (K given), gemm.hip:choose_cfg's otherwise."""
    x3 = _X3_TILE_MODEL and K is not None and min(M, N, K) >= _X3_MINDIM
    best, best_tiles = None, 1
    for bm, bn, pen in (_X3_CFGS if x3 else _TILE_CFGS):
        t = -(-M // bm) * -(-N // bn)
        cost = -(-t // 256) * bm * bn * pen
        if best is None or cost < best:
            best, best_tiles = cost, t
    return best_tiles


def gemm_tn(a, b):
    """out[M,N] = a[K,M]^T @ b[K,N]        (weight gradient: dY^T @ X, reduction over pixels)"""
    _act(a, b)          # both fp32, or both half (activation gradient x activation); the weight gradient is fp32
    K, M = a.shape
    N = b.shape[1]
    assert b.shape[0] == K
    split = _pick_split(_tiles(M, N, K), K)
    r = _tuned_launch("tn", a, b, None, M, N, K, M, N, 1, 1, False, None,          # fp32 or half operands, fp32 result
                      lambda scr: _gemm(a, b, scr, M, N, K, M, N, N, 1, 1, 2 if split > 1 else 0, split))
    if r is not None:
        return r
    if split > 1:
        out = split_out((M, N), a)
        return _gemm(a, b, out, M, N, K, M, N, N, 1, 1, 2, split)
    out = empty((M, N), a)
    return _gemm(a, b, out, M, N, K, M, N, N, 1, 1, 0, 1)


def conv_geom(N, Hin, Win, Cin, Hout, Wout, KH, KW, stride, pad_t, pad_l, transposed=0):
    g = ConvGeom()
    g.N, g.Hin, g.Win, g.Cin, g.Hout, g.Wout = N, Hin, Win, Cin, Hout, Wout
    g.KH, g.KW, g.stride, g.pad_t, g.pad_l, g.transposed = KH, KW, stride, pad_t, pad_l, transposed
    return g


_CONV_SPLITK = True


_CONV_SMALL = True
_CONV_SMALL_MIN_M = 262144          # one thread per output pixel: below ~1 wave per SIMD x 4 the GEMM path wins
_CONV_SMALL_OK = {}


def _conv_small_supported(Ci, Co, KH, KW):
    key = (Ci, Co, KH, KW)
    if key not in _CONV_SMALL_OK:
        _CONV_SMALL_OK[key] = _call("ud_conv_small_supported", Ci, Co, KH, KW) == 1
    return _CONV_SMALL_OK[key]


_CONV_SMALL_WS = {}
_CONV_SMALL_WGRAD_OK = {}


def _conv_small_wgrad_supported(Ci, Ma, KH, KW):
    key = (Ci, Ma, KH, KW)
    if key not in _CONV_SMALL_WGRAD_OK:
        _CONV_SMALL_WGRAD_OK[key] = _call("ud_conv_small_wgrad_supported", Ci, Ma, KH, KW) == 1
    return _CONV_SMALL_WGRAD_OK[key]


def conv_gather_nt(x, wmat, g):
    """Implicit-GEMM conv: rows (n,oh,ow) x k=(tap,ci) gathered from x[N,Hin,Win,Cin]; wmat[Cout, KH*KW*Cin].
    Returns [N, Hout, Wout, Cout]."""
    _chk(x, wmat)
    M = g.N * g.Hout * g.Wout
    K = g.KH * g.KW * g.Cin
    Co = wmat.shape[0]
    assert wmat.shape[1] == K and x.numel() == g.N * g.Hin * g.Win * g.Cin
    if _CONV_SMALL and M >= _CONV_SMALL_MIN_M and _conv_small_supported(g.Cin, Co, g.KH, g.KW):
        # few channels at image resolution (decoder tail, stem): direct conv, one thread per output pixel
        out = empty((g.N, g.Hout, g.Wout, Co), x)
        _call("ud_conv_small", C.byref(g), _p(x), _p(wmat), _p(out), Co, _stream())
        return out
    # split-K for the under-filled launches (e.g. the 3x3 filter conv at 8x8: M = 2048, K = 2448 -> 80 tiles);
    # an A/B inside one gpurun call decides (UD_CONV_SPLITK=0 disables)
    split = _fwd_split(M, Co, K) if _CONV_SPLITK else 1
    if g.Cin % 4 == 0:                                  # the shapes the BF16-pipe gather takes: tuned like the plain GEMMs
        gk = (g.Hin, g.Win, g.Cin, g.KH, g.KW, g.stride, g.transposed)
        tmp = []

        def scratch():
            if not tmp:
                tmp.append(empty((M, Co), x))
            return tmp[0]
        tuned = _tuned_plan(("conv",) + gk, M, Co, K,
                            lambda cfg, sp: _gemm(x, wmat, scratch(), M, Co, K, 0, K, Co, 2, 0, 2 if sp > 1 else 0, sp, geom=g, cfg=cfg),
                            lambda: _gemm(x, wmat, scratch(), M, Co, K, 0, K, Co, 2, 0, 2 if split > 1 else 0, split, geom=g))
        if tuned is not None:
            cfg, sp = tuned
            out = (split_out if sp > 1 else empty)((g.N, g.Hout, g.Wout, Co), x)
            _gemm(x, wmat, out, M, Co, K, 0, K, Co, 2, 0, 2 if sp > 1 else 0, sp, geom=g, cfg=cfg)
            return out
    if split > 1:
        out = split_out((g.N, g.Hout, g.Wout, Co), x)
        _gemm(x, wmat, out, M, Co, K, 0, K, Co, 2, 0, 2, split, geom=g)
        return out
    out = empty((g.N, g.Hout, g.Wout, Co), x)
    _gemm(x, wmat, out, M, Co, K, 0, K, Co, 2, 0, geom=g)
    return out


def conv_gather_wgrad(a, x, g):
    """out[Ma, KH*KW*Cin] = a[(n,oh,ow), Ma]^T @ gather(x)[(n,oh,ow), (tap,ci)]."""
    _chk(a, x)
    Kdim = g.N * g.Hout * g.Wout
    Ma = a.shape[-1]
    assert a.numel() == Kdim * Ma
    Ncols = g.KH * g.KW * g.Cin
    if _CONV_SMALL and Kdim >= _CONV_SMALL_MIN_M and _conv_small_wgrad_supported(g.Cin, Ma, g.KH, g.KW):
        # a 20 x 180 (3 x 180, 48 x 27) result reduced over 524 288 rows: streamed through LDS, register-blocked
        need = _call("ud_conv_small_wgrad_ws_floats", g.Cin, Ma)
        ws = _CONV_SMALL_WS.get(_key(a))
        if ws is None or ws.numel() < need:
            ws = _CONV_SMALL_WS[_key(a)] = empty((need,), a)
        out = empty((Ma, Ncols), a)
        _call("ud_conv_small_wgrad", C.byref(g), _p(a), _p(x), _p(ws), _p(out), Ma, _stream())
        return out
    split = _pick_split(_tiles(Ma, Ncols), Kdim)
    if g.Cin % 4 == 0 and Ma % 4 == 0:
        gk = (g.Hin, g.Win, g.Cin, g.KH, g.KW, g.stride, g.transposed)
        tmp = []

        def scratch():
            if not tmp:
                tmp.append(empty((Ma, Ncols), a))
            return tmp[0]
        tuned = _tuned_plan(("convw",) + gk, Ma, Ncols, Kdim,
                            lambda cfg, sp: _gemm(a, x, scratch(), Ma, Ncols, Kdim, Ma, 0, Ncols, 1, 2, 2 if sp > 1 else 0, sp, geom=g, cfg=cfg),
                            lambda: _gemm(a, x, scratch(), Ma, Ncols, Kdim, Ma, 0, Ncols, 1, 2, 2 if split > 1 else 0, split, geom=g))
        if tuned is not None:
            cfg, sp = tuned
            out = (split_out if sp > 1 else empty)((Ma, Ncols), a)
            return _gemm(a, x, out, Ma, Ncols, Kdim, Ma, 0, Ncols, 1, 2, 2 if sp > 1 else 0, sp, geom=g, cfg=cfg)
    if split > 1:
        out = split_out((Ma, Ncols), a)
        return _gemm(a, x, out, Ma, Ncols, Kdim, Ma, 0, Ncols, 1, 2, 2, split, geom=g)
    out = empty((Ma, Ncols), a)
    return _gemm(a, x, out, Ma, Ncols, Kdim, Ma, 0, Ncols, 1, 2, 0, 1, geom=g)


# ---------------------------------------------------------------------------------------------
# normalisation / column reductions
# ---------------------------------------------------------------------------------------------
_REDUCE_WS = {}          # device -> fp64 scratch shared by every column reduction
_REDUCE_WS_MIN = 1 << 20


def _reduce_ws(ref, G, R, Cc):
    """Scratch for the per-workgroup partial sums (include/unidefense_hip.h): a reduction's finalize launch has
    consumed it before the next reduction on the stream starts, so one buffer per device serves all calls."""
    need = _call("ud_reduce_ws_doubles", G, R, Cc)
    ws = _REDUCE_WS.get(_key(ref))
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, _REDUCE_WS_MIN), dtype=torch.float64, device=ref.device)
        _REDUCE_WS[_key(ref)] = ws
    return ws


def norm_stats(x2, G, R, eps, momentum=0.0, running_mean=None, running_var=None):
    """x2: [G*R, C].  Returns (mean[G,C], invstd[G,C]); updates running stats in place when given."""
    _chk(x2)
    Cc = x2.shape[-1]
    mean = empty((G, Cc), x2)
    invstd = empty((G, Cc), x2)
    _call("ud_norm_stats", _p(x2), G, R, Cc, eps, _p(_reduce_ws(x2, G, R, Cc)), _p(mean), _p(invstd), None,
          momentum, _p(running_mean), _p(running_var), _stream())
    return mean, invstd


def norm_stats_local(x2, G, R, eps):
    """mv[2, G, C] = (mean, biased var) per (g, c) — the per-rank half of a SyncBatchNorm forward, laid out as the
    all_gather payload."""
    _chk(x2)
    Cc = x2.shape[-1]
    mv = empty((2, G, Cc), x2)
    invstd = empty((G, Cc), x2)
    _call("ud_norm_stats", _p(x2), G, R, Cc, eps, _p(_reduce_ws(x2, G, R, Cc)), _p(mv[0]), _p(invstd), _p(mv[1]),
          0.0, None, None, _stream())
    return mv


def syncbn_combine(gathered, world, Cc, rows_per_rank, eps, momentum, running_mean, running_var):
    """gathered [world, 2, C] -> (mean[1,C], invstd[1,C]); running statistics updated in place when given."""
    _chk(gathered)
    mean = empty((1, Cc), gathered)
    invstd = empty((1, Cc), gathered)
    _call("ud_syncbn_combine", _p(gathered), world, Cc, rows_per_rank, eps, momentum, _p(running_mean),
          _p(running_var), _p(mean), _p(invstd), _stream())
    return mean, invstd


def norm_bwd_sums(x2, dy, G, R, mean, invstd, gamma, beta, act):
    """Reductions of the norm backward only: returns (s[2,G,C], dgamma[C], dbeta[C])."""
    _chk(x2, dy)
    Cc = x2.shape[-1]
    s = empty((2, G, Cc), x2)
    dg = empty((Cc,), x2)
    db = empty((Cc,), x2)
    _call("ud_norm_bwd", _p(x2), _p(dy), G, R, Cc, _p(mean), _p(invstd), _p(gamma), _p(beta), int(act),
          _p(_reduce_ws(x2, G, R, Cc)), _p(s[0]), _p(s[1]), _p(dg), _p(db), None, _stream())
    return s, dg, db


def norm_bwd_apply(x2, dy, G, R, mean, invstd, gamma, beta, s, inv_count, act):
    _chk(x2, dy, s)
    dx = torch.empty_like(x2)
    _call("ud_norm_bwd_apply", _p(x2), _p(dy), G, R, x2.shape[-1], _p(mean), _p(invstd), _p(gamma), _p(beta),
          _p(s[0]), _p(s[1]), inv_count, int(act), _p(dx), _stream())
    return dx


def norm_apply(x2, G, R, mean, invstd, gamma, beta, act):
    _chk(x2, mean, invstd, gamma, beta)
    y = torch.empty_like(x2)
    _call("ud_norm_apply_fwd", _p(x2), G, R, x2.shape[-1], _p(mean), _p(invstd), _p(gamma), _p(beta), int(act),
          _p(y), _stream())
    return y


def norm_bwd(x2, dy, G, R, mean, invstd, gamma, beta, act):
    """Returns (dx, dgamma[C], dbeta[C])."""
    _chk(x2, dy)
    Cc = x2.shape[-1]
    s = empty((2, G, Cc), x2)
    dg = empty((Cc,), x2)
    db = empty((Cc,), x2)
    dx = torch.empty_like(x2)
    _call("ud_norm_bwd", _p(x2), _p(dy), G, R, Cc, _p(mean), _p(invstd), _p(gamma), _p(beta), int(act),
          _p(_reduce_ws(x2, G, R, Cc)), _p(s[0]), _p(s[1]), _p(dg), _p(db), _p(dx), _stream())
    return dx, dg, db


# One-launch normalisation (csrc/norm.hip: norm_fwd_fused / norm_bwd_fused): statistics + apply in ONE kernel whose workgroups
# exchange their partial sums through agent-scope atomics (no fence).  Built to cut the launch-bound InstanceNorms of the decoder
# and the ResNet variants' BatchNorms from 3 / 4-5 launches to 1 / 1-2 — correct (tests/test_a_kernels_gpu.py), and measured
# SLOWER on this part: the exchange is four dependent round trips to the memory side (publish, count, poll, read: ~17 us per
# kernel) against two ~4.5 us launches — UDEB4 bs 32 25.07 -> 25.51 ms, UDR50 320^2 22.6 -> 25.2 (its 50 MB tensors also want
# 2000 workgroups, the one-launch form runs <= 768 resident ones), UDR18 4.66 -> 5.00 (profiles/r06/norm_one_launch_ab.txt).
# OFF; kept as the measured answer to "exchange inside the kernel instead of a launch".  A/B: tools/run_with.py kernels._NORM_FUSED=True
_NORM_FUSED = False
_NORM_FUSED_MAX_BYTES = 64 << 20


def norm_fused_ok(x2, G, R):
    Cc = x2.shape[-1]
    return (_NORM_FUSED and norm_fused_takes(x2, G, R))


def norm_fused_takes(x2, G, R):
    Cc = x2.shape[-1]
    return (x2.is_cuda and x2.dtype == torch.float32 and Cc % 4 == 0 and Cc >= 4
            and x2.numel() * 4 <= _NORM_FUSED_MAX_BYTES and G * R == x2.shape[0])


def _norm_fused_ws(x2, G, R):
    Cc = x2.shape[-1]
    slots = _ws64_t(x2, _call("ud_norm_fused_ws_doubles", G, R, Cc))
    counters = zeros((_call("ud_norm_fused_counters", G, R, Cc),), x2)          # (zero words: 32-bit counters)
    return slots, counters


def norm_fwd_fused(x2, G, R, gamma, beta, act, eps, momentum=0.0, running_mean=None, running_var=None):
    """(y, mean[G,C], invstd[G,C]) = ud_norm_stats + ud_norm_apply_fwd in one launch; running statistics moved when given (G = 1)"""
    _chk(x2, gamma, beta)
    Cc = x2.shape[-1]
    mean, invstd, y = empty((G, Cc), x2), empty((G, Cc), x2), torch.empty_like(x2)
    slots, counters = _norm_fused_ws(x2, G, R)
    _call("ud_norm_fwd_fused", _p(x2), G, R, Cc, _p(gamma), _p(beta), int(act), eps, _p(slots), _p(counters), _p(mean), _p(invstd),
          momentum, _p(running_mean), _p(running_var), _p(y), _stream())
    return y, mean, invstd


def norm_bwd_fused(x2, dy, G, R, mean, invstd, gamma, beta, act):
    """(dx, dgamma[C], dbeta[C]) = ud_norm_bwd in one launch (G > 1: + the group sum of dgamma / dbeta)"""
    _chk(x2, dy)
    Cc = x2.shape[-1]
    s = empty((2, G, Cc), x2)
    dg, db, dx = empty((Cc,), x2), empty((Cc,), x2), torch.empty_like(x2)
    slots, counters = _norm_fused_ws(x2, G, R)
    _call("ud_norm_bwd_fused", _p(x2), _p(dy), G, R, Cc, _p(mean), _p(invstd), _p(gamma), _p(beta), int(act), _p(slots),
          _p(counters), _p(s[0]), _p(s[1]), _p(dg), _p(db), _p(dx), _stream())
    return dx, dg, db


def group_colsum(x2, G, R, scale):
    _chk(x2)
    Cc = x2.shape[-1]
    out = empty((G, Cc), x2)
    _call("ud_group_colsum", _p(x2), G, R, Cc, scale, _p(_reduce_ws(x2, G, R, Cc)), _p(out), _stream())
    return out


def group_coldot(a2, b2, G, R, scale=1.0):
    _chk(a2, b2)
    Cc = a2.shape[-1]
    out = empty((G, Cc), a2)
    _call("ud_group_coldot", _p(a2), _p(b2), G, R, Cc, scale, _p(_reduce_ws(a2, G, R, Cc)), _p(out), _stream())
    return out


def bcast_rows(g, HW, scale):
    """g[N,C] -> out[N,HW,C] = g * scale."""
    _chk(g)
    N, Cc = g.shape
    out = empty((N, HW, Cc), g)
    _call("ud_bcast_rows", _p(g), scale, _p(out), N, HW, Cc, _stream())
    return out


# ---------------------------------------------------------------------------------------------
# depthwise conv
# ---------------------------------------------------------------------------------------------
def dwconv_fwd(x, wt, K, stride, pad_t, pad_l, Ho, Wo):
    h = _act(x)
    _chk(wt)
    N, H, W, Cc = x.shape
    y = empty((N, Ho, Wo, Cc), x, x.dtype)
    _call("ud_dwconv_fwd", _p(x), _p(wt), _p(y), N, H, W, Cc, Ho, Wo, K, stride, pad_t, pad_l, h, _stream())
    return y


_DW_WT_TABLES = {}


def dw_weights_tapmajor(weights):
    """[C,1,k,k] depthwise parameters -> {id(w): wt[k*k][C]} (views of one flat buffer), one launch for all of them.
    The pointer table lives on the device and is rebuilt only when a parameter's storage moved."""
    if not weights:
        return {}
    _chk(*weights)
    key = tuple(w.data_ptr() for w in weights)
    ent = _DW_WT_TABLES.get(id(weights[0]))
    if ent is None or ent[0] != key:
        rows, off = [], 0
        for w in weights:
            Cc, kk = w.shape[0], w.shape[-1] * w.shape[-2]
            rows.append([w.data_ptr(), Cc, kk, off])
            off += (Cc * kk + 63) // 64 * 64
        table = torch.tensor(rows, dtype=torch.int64).to(weights[0].device)
        ent = _DW_WT_TABLES[id(weights[0])] = (key, table, rows, off, max(r[1] * r[2] for r in rows))
    _, table, rows, total, max_elems = ent
    flat = empty((total,), weights[0])
    _call("ud_dw_weights_tapmajor", _p(table), len(rows), max_elems, _p(flat), _stream())
    return {id(w): flat[r[3]:r[3] + r[1] * r[2]].view(r[2], r[1]) for w, r in zip(weights, rows)}


def dwconv_bwd_data(dy, wt, K, stride, pad_t, pad_l, H, W, add=None):
    """add: another contribution to the same input gradient, summed in the store (saves an axpby pass)."""
    h = _act(dy, add)
    _chk(wt)
    N, Ho, Wo, Cc = dy.shape
    dx = empty((N, H, W, Cc), dy, dy.dtype)
    assert add is None or add.shape == dx.shape
    _call("ud_dwconv_bwd_data", _p(dy), _p(wt), _p(add), _p(dx), N, H, W, Cc, Ho, Wo, K, stride, pad_t, pad_l, h,
          _stream())
    return dx


_DW_WGRAD_THREADS = 131072


def dwconv_bwd_weight(x, dy, K, stride, pad_t, pad_l):
    _chk(x, dy)
    N, H, W, Cc = x.shape
    _, Ho, Wo, _ = dy.shape
    # one thread per (channel, group of output rows): aim at >= ~128k threads, at most one group per row
    chunks = max(1, min(N * Ho, -(-_DW_WGRAD_THREADS // Cc)))
    parts = _call("ud_dwconv_bwd_weight_parts", Cc, chunks)
    part = empty((parts, K * K, Cc), x)
    dwt = empty((Cc, K * K), x)        # the parameter's own layout (weight [C,1,K,K])
    _call("ud_dwconv_bwd_weight", _p(x), _p(dy), _p(dwt), _p(part), chunks, N, H, W, Cc, Ho, Wo, K, stride, pad_t,
          pad_l, _stream())
    return dwt


# ---------------------------------------------------------------------------------------------
# FFT
# ---------------------------------------------------------------------------------------------
# Two-pass form of the 32 x 32 / 64 x 64 transforms (csrc/fft.hip: a row kernel and a column kernel with the half-spectrum
# between them in HBM, whole lines per access) where it beats the LDS-resident kernel — measured per (variant, size,
# storage) with tools/bench_fft2p.py (profiles/r03/fft_two_pass.txt): half storage, where the one-kernel form reads 16 / 32
# bytes per pixel, everything at 64 x 64 (1.3 - 3.7x) and the plain inverse at 32 x 32 (1.7 - 2x); fp32 storage the two fused
# 64 x 64 variants (1.1 - 1.7x).  _FFT_TWO_PASS: None = this table, True / False = everywhere possible / nowhere (tests).
_FFT_TWO_PASS = None
_FFT2P_POLICY = {("rfft", 64, True), ("rfft_ex", 64, True), ("irfft", 64, True), ("irfft_mix", 64, True),
                 ("irfft", 32, True),
                 ("rfft_ex", 64, False), ("irfft_mix", 64, False)}


def _fft_two_pass(kind, S, half):
    if S not in (32, 64):
        return False
    if _FFT_TWO_PASS is not None:
        return _FFT_TWO_PASS
    return (kind, S, bool(half)) in _FFT2P_POLICY


def _fft_ws(N, S, Cc, like):
    return empty((N, S, S // 2 + 1, 2 * Cc), like, torch.float32)


_FFT_KERNEL_SIZES = (8, 16, 32, 64, 10, 20, 40, 80, 12, 24, 48)          # csrc/fft.hip: 2^k, 5 * 2^k, 3 * 2^k in registers


def fft_kernel_size(S):
    """True where csrc/fft.hip transforms S x S maps in registers; other sides (95 = 5 * 19 of the EfficientNet-b4 trunk at
    380 x 380) go through DFT matrices on the GEMM kernels (_rfft2_generic / _irfft2_generic) — correct, not fast"""
    return S in _FFT_KERNEL_SIZES


_FFT_COLW = {}


def _fft_col_weights(S, w_interior, like, twice):
    """per-kx factors of the half spectrum: 1 on the self-conjugate columns (kx = 0, and S/2 for even S), w_interior elsewhere
    (x 2 for the inverse: the Hermitian extension the kernels make explicitly).  Built on the host once per (S, w, device): an
    element assignment on a device tensor is a synchronous copy, which a graph capture refuses."""
    key = (S, float(w_interior), bool(twice), str(like.device))
    f = _FFT_COLW.get(key)
    if f is None:
        Wh = S // 2 + 1
        h = torch.full((Wh,), float(w_interior) * (2.0 if twice else 1.0))
        h[0] = 1.0
        if S % 2 == 0:
            h[Wh - 1] = 1.0
        f = _FFT_COLW[key] = h.to(like.device)
    return f


_DFT_PIXEL_CACHE = {}


def _ceil4(v):
    return -(-v // 4) * 4


def _dft_pixel_mats(S, scale, w_interior, inverse, device):
    """The two matrices of a pixel-major 2-D real DFT done as two batched GEMMs on the tensor AS IT LIES ([N, S, S, C] <->
    [N, S, S/2+1, 2C]; no plane copies), stored k-major [K][ceil4(M)] (ud_gemm a_mode 1: no condition on the reduction length, the
    slack columns zero) — float64 on the host, once per (S, scale, w, direction, device):
      forward   pass 1, per image, along h:      T[(ky, ri), (w, c)]  = sum_h   A1[(ky, ri), h]      x[h, (w, c)]
                pass 2, per (image, ky), along w: Y[(kx, ri), c]       = sum_{ri', w} A2[(kx, ri), (ri', w)] T[(ri', w), c]
                (scale and the half spectrum's column weights folded into A2)
      inverse   pass 1, per (image, ky), along kx: U[(ri', w), c]      = sum_{kx, ri} A3[(ri', w), (kx, ri)] Y[(kx, ri), c]
                (Hermitian multiplicity x column weight folded in; U complex: x = Re sum_ky e^{+i theta} U)
                pass 2, per image, along ky:      x[h, (w, c)]         = sum_{ky, ri'} A4[h, (ky, ri')]    U[(ky, ri'), (w, c)]"""
    key = (S, float(scale), float(w_interior), bool(inverse), str(device))
    m = _DFT_PIXEL_CACHE.get(key)
    if m is not None:
        return m
    Wh = S // 2 + 1
    k = torch.arange(S, dtype=torch.float64)
    ang = 2.0 * math.pi * torch.outer(k, k) / S                     # [k][position]
    cs, sn = torch.cos(ang), torch.sin(ang)
    selfconj = torch.zeros(Wh, dtype=torch.bool)
    selfconj[0] = True
    if S % 2 == 0:
        selfconj[Wh - 1] = True
    colw = torch.where(selfconj, torch.ones(Wh, dtype=torch.float64), torch.full((Wh,), float(w_interior), dtype=torch.float64))
    if not inverse:
        a1 = torch.zeros(S, _ceil4(2 * S), dtype=torch.float64)            # [h][(ky, ri)]
        a1[:, 0:2 * S:2] = cs.t()                                          # e^{-i theta}: Re = cos, Im = -sin
        a1[:, 1:2 * S:2] = -sn.t()
        g = colw * float(scale)
        a2 = torch.zeros(2 * S, _ceil4(2 * Wh), dtype=torch.float64)       # [(ri', w)][(kx, ri)]
        c2, s2 = (cs[:Wh] * g[:, None]).t(), (sn[:Wh] * g[:, None]).t()    # [w][kx]
        a2[:S, 0:2 * Wh:2], a2[S:, 0:2 * Wh:2] = c2, s2                    # Re Y = g (Tr cos + Ti sin)
        a2[:S, 1:2 * Wh:2], a2[S:, 1:2 * Wh:2] = -s2, c2                   # Im Y = g (Ti cos - Tr sin)
        mats = (a1, a2)
    else:
        mf = colw * torch.where(selfconj, 1.0, 2.0)
        a3 = torch.zeros(2 * Wh, _ceil4(2 * S), dtype=torch.float64)       # [(kx, ri)][(ri', w)]
        c3, s3 = cs[:Wh] * mf[:, None], sn[:Wh] * mf[:, None]              # [kx][w]
        a3[0::2, :S], a3[1::2, :S] = c3, -s3                               # Re U = mf (Yr cos - Yi sin)
        a3[0::2, S:2 * S], a3[1::2, S:2 * S] = s3, c3                      # Im U = mf (Yr sin + Yi cos)
        a4 = torch.zeros(2 * S, _ceil4(S), dtype=torch.float64)            # [(ky, ri')][h]
        a4[0::2, :S] = cs * float(scale)                                   # x = scale (Ur cos - Ui sin)
        a4[1::2, :S] = -sn * float(scale)
        mats = (a3, a4)
    m = _DFT_PIXEL_CACHE[key] = tuple(t.to(torch.float32).to(device).contiguous() for t in mats)
    return m


_FFT_GENERIC_PIXEL = True          # A/B: tools/run_with.py kernels._FFT_GENERIC_PIXEL=False (the plane-copy form)


def _rfft2_generic(x, scale, w_interior):
    """rfft2 of ud_rfft2's contract for any side (95 = 5 * 19 of the 380 x 380 trunk): two batched GEMMs against DFT matrices on
    the pixel-major tensor itself (_dft_pixel_mats) — no plane copies, no torch kernel"""
    N, S, _, Cc = x.shape
    if not (_FFT_GENERIC_PIXEL and Cc % 4 == 0):
        return _rfft2_generic_planes(x, scale, w_interior)
    Wh = S // 2 + 1
    xs = x if x.dtype == torch.float32 else x.float()
    a1, a2 = _dft_pixel_mats(S, scale, w_interior, False, x.device)
    T = empty((N, 2 * S, S, Cc), xs)
    _gemm(a1, xs, T, 2 * S, S * Cc, S, a1.shape[1], S * Cc, S * Cc, 1, 1, 0, batch=N, strideA=0, strideB=S * S * Cc,
          strideC=2 * S * S * Cc)
    Y = empty((N, S, Wh, 2 * Cc), xs)
    _gemm(a2, T, Y, 2 * Wh, Cc, 2 * S, a2.shape[1], Cc, Cc, 1, 1, 0, batch=N * S, strideA=0, strideB=2 * S * Cc,
          strideC=2 * Wh * Cc)
    return Y if x.dtype == torch.float32 else Y.to(x.dtype)


def _irfft2_generic(Y, scale, w_interior):
    """irfft2 of ud_irfft2's contract for any side: x = scale * C2R(f Y), as two batched GEMMs on the pixel-major tensors"""
    N, S, Wh, C2 = Y.shape
    Cc = C2 // 2
    if not (_FFT_GENERIC_PIXEL and Cc % 4 == 0):
        return _irfft2_generic_planes(Y, scale, w_interior)
    Ys = Y if Y.dtype == torch.float32 else Y.float()
    a3, a4 = _dft_pixel_mats(S, scale, w_interior, True, Y.device)
    U = empty((N, S, 2 * S, Cc), Ys)                       # [n][ky][(ri', w)][c]
    _gemm(a3, Ys, U, 2 * S, Cc, 2 * Wh, a3.shape[1], Cc, Cc, 1, 1, 0, batch=N * S, strideA=0, strideB=2 * Wh * Cc,
          strideC=2 * S * Cc)
    x = empty((N, S, S, Cc), Ys)
    _gemm(a4, U, x, S, S * Cc, 2 * S, a4.shape[1], S * Cc, S * Cc, 1, 1, 0, batch=N, strideA=0, strideB=2 * S * S * Cc,
          strideC=S * S * Cc)
    return x if Y.dtype == torch.float32 else x.to(Y.dtype)


def _rfft2_generic_planes(x, scale, w_interior):
    """the plane-copy form (channel counts that are not multiples of 4): pixel-major -> (padded) planes in one strided copy, three
    DFT-matrix GEMMs (_dft_planes_fwd), planes -> pixel-major with the column weights in one pass"""
    N, S, _, Cc = x.shape
    Wh = S // 2 + 1
    Sp = -(-S // 4) * 4
    dp = torch.zeros((N, Cc, Sp, Sp), device=x.device) if Sp != S else torch.empty((N, Cc, S, S), device=x.device)
    dp[:, :, :S, :S] = x.permute(0, 3, 1, 2)
    Yp = _dft_planes_fwd(dp.view(N * Cc, Sp, Sp), S, ortho=False)             # [P, 2S, Whp]: rows Re(ky) | Im(ky)
    Y = torch.empty((N, S, Wh, 2, Cc), device=x.device, dtype=x.dtype)
    torch.mul(Yp.view(N, Cc, 2, S, Yp.shape[-1])[..., :Wh].permute(0, 3, 4, 2, 1),
              (_fft_col_weights(S, w_interior, x, False) * float(scale)).view(1, 1, Wh, 1, 1), out=Y)
    return Y.view(N, S, Wh, 2 * Cc)


def _irfft2_generic_planes(Y, scale, w_interior):
    """x = scale * F^T(m f Y), F the unnormalised rfft2 (its adjoint on the GEMM kernels: _dft_planes_adj), m the Hermitian
    multiplicity, f = w_interior off the self-conjugate columns"""
    N, S, Wh, C2 = Y.shape
    Cc = C2 // 2
    Whp = -(-Wh // 4) * 4
    dY = torch.zeros((N, Cc, 2, S, Whp), device=Y.device) if Whp != Wh else torch.empty((N, Cc, 2, S, Whp), device=Y.device)
    torch.mul(Y.view(N, S, Wh, 2, Cc).permute(0, 4, 3, 1, 2),
              (_fft_col_weights(S, w_interior, Y, True) * float(scale)).view(1, 1, 1, 1, Wh), out=dY[..., :Wh])
    xp = _dft_planes_adj(dY.view(N * Cc, 2 * S, Whp), S, ortho=False)          # [P, Sp, Sp], the transform in [:S, :S]
    x = torch.empty((N, S, S, Cc), device=Y.device, dtype=Y.dtype)
    x.copy_(xp.view(N, Cc, xp.shape[-2], xp.shape[-1])[:, :, :S, :S].permute(0, 2, 3, 1))
    return x


def rfft2(x, scale, w_interior=1.0, want_absmax=False):
    """x[N,S,S,C] -> Y[N,S,S/2+1,2C] (Re | Im channel halves).  want_absmax: Y._ud_absmax = 256 slots of |Y|max."""
    if not fft_kernel_size(x.shape[1]):
        return _rfft2_generic(x, scale, w_interior)
    if want_absmax and amax_slots(x) is not None:
        return rfft2_ex(x, scale, w_interior, want_absmax=True)[0]
    h = _act(x)
    N, S, S2, Cc = x.shape
    assert S == S2
    Y = empty((N, S, S // 2 + 1, 2 * Cc), x, x.dtype)
    if _fft_two_pass("rfft", S, h):
        _call("ud_rfft2_two_pass", _p(x), _p(Y), _p(_fft_ws(N, S, Cc, x)), N, S, Cc, scale, w_interior, None, None, None, 0,
              None, None, h, None, _stream())
        return Y
    _call("ud_rfft2", _p(x), _p(Y), N, S, Cc, scale, w_interior, h, _stream())
    return Y


def irfft2(Y, scale, w_interior=1.0):
    """Y[N,S,S/2+1,2C] -> x[N,S,S,C]."""
    h = _act(Y)
    N, S, Wh, C2 = Y.shape
    assert Wh == S // 2 + 1 and C2 % 2 == 0
    if not fft_kernel_size(S):
        return _irfft2_generic(Y, scale, w_interior)
    x = empty((N, S, S, C2 // 2), Y, Y.dtype)
    if _fft_two_pass("irfft", S, h):
        _call("ud_irfft2_two_pass", _p(Y), _p(x), _p(_fft_ws(N, S, C2 // 2, Y)), N, S, C2 // 2, scale, w_interior, None, None,
              None, None, None, h, _stream())
        return x
    _call("ud_irfft2", _p(Y), _p(x), N, S, C2 // 2, scale, w_interior, h, _stream())
    return x


# ---------------------------------------------------------------------------------------------
# small FC, SE, mixing, elementwise
# ---------------------------------------------------------------------------------------------
def fc_fwd(x, W, b, act_in=0):
    _chk(x, W, b)
    N, I = x.shape
    O = W.shape[0]
    y = empty((N, O), x)
    _call("ud_fc_fwd", _p(x), _p(W), _p(b), _p(y), N, I, O, act_in, _stream())
    return y


def fc_bwd(dy, W, x, act_in=0, need_dx=True, need_db=True):
    _chk(dy, W, x)
    N, I = x.shape
    O = W.shape[0]
    dx = empty((N, I), x) if need_dx else None
    dW = empty((O, I), x)
    db = empty((O,), x) if need_db else None
    _call("ud_fc_bwd", _p(dy), _p(W), _p(x), _p(dx), _p(dW), _p(db), N, I, O, act_in, _stream())
    return dx, dW, db


def se_scale_fwd(x, s):
    _chk(x, s)
    N, H, W, Cc = x.shape
    y = torch.empty_like(x)
    _call("ud_se_scale_fwd", _p(x), _p(s), _p(y), N, H * W, Cc, _stream())
    return y


def se_scale_bwd(dy, s, dpool):
    _chk(dy, s, dpool)
    N, H, W, Cc = dy.shape
    dx = torch.empty_like(dy)
    _call("ud_se_scale_bwd", _p(dy), _p(s), _p(dpool), _p(dx), N, H * W, Cc, _stream())
    return dx


def sigmoid_grad_mul_(s, v):
    _chk(s, v)
    _call("ud_sigmoid_grad_mul", _p(s), _p(v), v.numel(), _stream())
    return v


def sfmix_fwd(spat, freq, alpha, pool):
    h = _act(spat, freq)
    _chk(alpha)
    N, Ho, Wo, Cc = spat.shape
    y = torch.empty_like(spat)
    _call("ud_sfmix_fwd", _p(spat), _p(freq), _p(alpha), _p(y), N, Ho, Wo, Cc, int(pool), h, _stream())
    return y


def sfmix_bwd(spat, freq, alpha, dy, pool):
    h = _act(spat, freq, dy)
    _chk(alpha)
    N, Ho, Wo, Cc = spat.shape
    nb = _call("ud_sfmix_blocks", N, Ho, Wo, Cc)
    part = torch.empty((nb,), dtype=torch.float64, device=spat.device)
    dspat = torch.empty_like(spat)
    dfreq = torch.empty_like(freq)
    dalpha = empty((), spat)
    _call("ud_sfmix_bwd", _p(spat), _p(freq), _p(alpha), _p(dy), _p(dspat), _p(dfreq), _p(part), _p(dalpha), N, Ho,
          Wo, Cc, int(pool), h, _stream())
    return dspat, dfreq, dalpha


def gate_mix_fwd(p, q, alpha):
    _chk(p, q, alpha)
    y = torch.empty_like(p)
    _call("ud_gate_mix_fwd", _p(p), _p(q), _p(alpha), _p(y), p.numel(), _stream())
    return y


def gate_mix_bwd(p, q, alpha, dy):
    _chk(p, q, alpha, dy)
    nb = _call("ud_gate_mix_blocks", p.numel())
    part = torch.empty((nb,), dtype=torch.float64, device=p.device)
    dp = torch.empty_like(p)
    dq = torch.empty_like(p)
    dalpha = empty((), p)
    _call("ud_gate_mix_bwd", _p(p), _p(q), _p(alpha), _p(dy), _p(dp), _p(dq), _p(part), _p(dalpha), p.numel(),
          _stream())
    return dp, dq, dalpha


def residual(x, skip, keep=None, inv_keep=1.0):
    """out = x * (keep[n] * inv_keep) + skip   (skip / keep may be None)."""
    _chk(x, skip, keep)
    out = torch.empty_like(x)
    _call("ud_residual_fwd", _p(x), _p(skip), _p(keep), inv_keep, _p(out), x.numel(), x.numel() // x.shape[0],
          _stream())
    return out


def axpby(a, alpha, b=None, beta=1.0, out=None):
    h = _act(a, b, out)
    if out is None:
        out = torch.empty_like(a)
    _call("ud_axpby", _p(a), alpha, _p(b), beta, _p(out), a.numel(), h, _stream())
    return out


def multi_add(dsts, srcs):
    """dsts[i] += srcs[i] for lists of contiguous fp32 device tensors of equal sizes, in ceil(n / 120) launches (the gradient
    accumulation of the train step's second backward: torch's AccumulateGrad launches once per parameter)"""
    n = len(dsts)
    if n == 0:
        return
    for d, s_ in zip(dsts, srcs):
        if not (d.is_cuda and s_.is_cuda and d.dtype == s_.dtype == torch.float32 and d.is_contiguous() and s_.is_contiguous()
                and d.numel() == s_.numel() and d.device == s_.device):
            raise ValueError("multi_add: contiguous float32 device tensors of equal sizes only")
    PA, LA = C.c_void_p * n, C.c_long * n
    _call("ud_multi_add", PA(*[d.data_ptr() for d in dsts]), PA(*[s_.data_ptr() for s_ in srcs]),
          LA(*[d.numel() for d in dsts]), n, _stream())


def mask_scale(x, mask, scale):
    _chk(x, mask)
    out = torch.empty_like(x)
    _call("ud_mask_scale", _p(x), _p(mask), scale, _p(out), x.numel(), _stream())
    return out


def absdiff(a, b=None):
    _chk(a, b)
    out = torch.empty_like(a)
    _call("ud_absdiff", _p(a), _p(b), _p(out), a.numel(), _stream())
    return out


def pix_to_planes(x, tanh=False):
    """[N,H,W,C] -> [N,C,H,W] (optionally through tanh)."""
    _chk(x)
    N, H, W, Cc = x.shape
    out = empty((N, Cc, H, W), x)
    _call("ud_pix_to_planes", _p(x), _p(out), N, Cc, H * W, 1 if tanh else 0, _stream())
    return out


def planes_to_pix(x, tanh_out=None):
    """[N,C,H,W] -> [N,H,W,C]; with tanh_out (planes): multiply by (1 - tanh_out^2)."""
    _chk(x, tanh_out)
    N, Cc, H, W = x.shape
    out = empty((N, H, W, Cc), x)
    _call("ud_planes_to_pix", _p(x), _p(tanh_out), _p(out), N, Cc, H * W, 2 if tanh_out is not None else 0,
          _stream())
    return out


def bilinear_fwd(x, Ho, Wo):
    _chk(x)
    N, Cc, Hi, Wi = x.shape
    y = empty((N, Cc, Ho, Wo), x)
    _call("ud_bilinear_fwd", _p(x), _p(y), N * Cc, Hi, Wi, Ho, Wo, _stream())
    return y


def bilinear_bwd(dy, Hi, Wi):
    _chk(dy)
    N, Cc, Ho, Wo = dy.shape
    dx = empty((N, Cc, Hi, Wi), dy)
    _call("ud_bilinear_bwd", _p(dy), _p(dx), N * Cc, Hi, Wi, Ho, Wo, _stream())
    return dx


def l1_fwd(a, b, scale):
    """out[n] = scale * sum |a[n] - b[n]|"""
    _chk(a, b)
    N = a.shape[0]
    per = a.numel() // N
    P = _call("ud_l1_chunks", per)
    part = empty((N, P), a)
    out = empty((N,), a)
    _call("ud_l1_fwd", _p(a), _p(b), _p(part), _p(out), N, per, scale, _stream())
    return out


def l1_bwd(a, b, g, scale, out=None):
    """da (+)= g[n] * scale * sign(a - b)   (accumulates into `out` when given)."""
    _chk(a, b, g, out)
    N = a.shape[0]
    acc = out is not None
    if out is None:
        out = torch.empty_like(a)
    _call("ud_l1_bwd", _p(a), _p(b), _p(g), scale, 1 if acc else 0, _p(out), N, a.numel() // N, _stream())
    return out


def dynfilter_fwd(proj2, diff2, w2, x2):
    _chk(proj2, diff2, w2, x2)
    M, Cc = proj2.shape
    D = diff2.shape[1]
    Cx = x2.shape[1]
    pre = empty((M, 2 + D), x2)
    argmax = torch.empty((M,), dtype=torch.int32, device=x2.device)
    mask = empty((M,), x2)
    out = torch.empty_like(x2)
    _call("ud_dynfilter_fwd", _p(proj2), _p(diff2), _p(w2), _p(x2), _p(pre), C.c_void_p(argmax.data_ptr()), _p(mask),
          _p(out), M, Cc, D, Cx, _stream())
    return out, mask, pre, argmax


def dynfilter_bwd(dout2, dmask_ext, x2, mask, argmax, w2, Cproj):
    _chk(dout2, dmask_ext, x2, mask, w2)
    M, Cx = x2.shape
    dx = torch.empty_like(x2)
    dlogit = empty((M,), x2)
    dproj = empty((M, Cproj), x2)
    _call("ud_dynfilter_bwd", _p(dout2), _p(dmask_ext), _p(x2), _p(mask), C.c_void_p(argmax.data_ptr()), _p(w2),
          _p(dx), _p(dlogit), _p(dproj), M, Cproj, Cx, _stream())
    return dx, dlogit, dproj


# ---------------------------------------------------------------------------------------------
# Large-plane rfft2 (S >= 128) of [P, S, S] planes as three batched MFMA GEMMs against DFT matrices
# (row transform, then the complex column transform as two real GEMMs).  Used for the 256x256 frequency
# reconstruction loss (model/unidefense.py:246-253).
# ---------------------------------------------------------------------------------------------
_DFT_CACHE = {}


def _dft_mats(S, device, ortho=True):
    """(fw_cos [Whp, Sp], fw_sin [Whp, Sp], fh [2S, 2 Sp]) with Sp = S rounded up to 4 and zero padding: every row stride stays a
    multiple of 16 bytes for odd sides too (95), so the products run on the matrix-pipe kernels, not the scalar-load fp32 one"""
    key = (S, str(device), ortho)
    if key in _DFT_CACHE:
        return _DFT_CACHE[key]
    Wh = S // 2 + 1
    Whp = -(-Wh // 4) * 4
    Sp = -(-S // 4) * 4
    s = 1.0 / math.sqrt(S) if ortho else 1.0
    k = torch.arange(S, dtype=torch.float64)
    ang = 2.0 * math.pi * torch.outer(k, k) / S          # [k][w]
    cosm, sinm = torch.cos(ang) * s, torch.sin(ang) * s
    fw_cos = torch.zeros(Whp, Sp, dtype=torch.float64)
    fw_sin = torch.zeros(Whp, Sp, dtype=torch.float64)
    fw_cos[:Wh, :S] = cosm[:Wh]
    fw_sin[:Wh, :S] = -sinm[:Wh]                           # Im of e^{-i t} = -sin
    # column transform on stacked [Re; Im]:  Yre = C Tre + S Tim ;  Yim = C Tim - S Tre
    fh = torch.zeros(2 * S, 2 * Sp, dtype=torch.float64)
    fh[:S, :S], fh[:S, Sp:Sp + S] = cosm, sinm
    fh[S:, :S], fh[S:, Sp:Sp + S] = -sinm, cosm
    mats = tuple(m.to(torch.float32).to(device).contiguous() for m in (fw_cos, fw_sin, fh))
    _DFT_CACHE[key] = mats
    return mats


_FFT_PLANES_WS = {}
_FFT_PLANES_SIZES = (128, 256, 320)          # csrc/fft_large.hip


def _fft_planes_ws(ref, P, S):
    need = _call("ud_rfft2_planes_ws_floats", P, S)
    ws = _FFT_PLANES_WS.get(_key(ref))
    if ws is None or ws.numel() < need:
        ws = _FFT_PLANES_WS[_key(ref)] = empty((need,), ref)
    return ws


def dft_rfft2_planes(d, ortho=True):
    """d: [P, S, S] real planes -> Y [P, 2S, Whp]: rows [0,S) = Re(ky), rows [S,2S) = Im(ky); columns
    [0, S/2] valid, the rest zero padding (Whp = ceil4(S/2+1)).  S in {128, 256, 320}: the row / column FFT kernels
    of csrc/fft_large.hip; other sizes: three batched GEMMs against DFT matrices."""
    _chk(d)
    P, S, _ = d.shape
    if S in _FFT_PLANES_SIZES:
        Y = empty((P, 2 * S, -(-(S // 2 + 1) // 4) * 4), d)
        _call("ud_rfft2_planes", _p(d), _p(Y), _p(_fft_planes_ws(d, P, S)), P, S, (1.0 / S) if ortho else 1.0, _stream())
        return Y
    Sp = -(-S // 4) * 4
    if Sp != S:                                     # rows and columns padded to a multiple of 4 floats (zeros)
        dp = torch.zeros((P, Sp, Sp), device=d.device)
        dp[:, :S, :S] = d
        d = dp
    return _dft_planes_fwd(d, S, ortho)


def _dft_planes_fwd(dp, S, ortho):
    """dp: [P, Sp, Sp] planes holding the S x S data in [:S, :S] and zeros around (Sp = S rounded up to 4: every stride AND
    every reduction length a multiple of 4 floats, the matrix-pipe kernels' condition) -> Y [P, 2S, Whp]"""
    P, Sp, _ = dp.shape
    fw_cos, fw_sin, fh = _dft_mats(S, dp.device, ortho)
    Whp = fw_cos.shape[0]
    d2 = dp.view(P * Sp, Sp)
    t_re = gemm_nt(d2, fw_cos)                      # [P*Sp, Whp]; rows >= S of a plane are zero
    t_im = gemm_nt(d2, fw_sin)
    Y = empty((P, 2 * S, Whp), dp)
    # Y_p = fh[:, :Sp] @ Tre_p + fh[:, Sp:] @ Tim_p     (batched over planes; A shared; the zero columns meet the zero rows)
    _gemm(fh, t_re, Y, 2 * S, Whp, Sp, 2 * Sp, Whp, Whp, 0, 1, 0, batch=P, strideA=0, strideB=Sp * Whp,
          strideC=2 * S * Whp)
    _gemm(fh, t_im, Y, 2 * S, Whp, Sp, 2 * Sp, Whp, Whp, 0, 1, 1, batch=P, strideA=0, strideB=Sp * Whp,
          strideC=2 * S * Whp, a_off=Sp)
    return Y


def dft_rfft2_planes_adjoint(dY, S, ortho=True):
    """Adjoint of dft_rfft2_planes: dY [P, 2S, Whp] -> dd [P, S, S]."""
    _chk(dY)
    P = dY.shape[0]
    if S in _FFT_PLANES_SIZES:
        dd = empty((P, S, S), dY)
        _call("ud_rfft2_planes_adjoint", _p(dY), _p(dd), _p(_fft_planes_ws(dY, P, S)), P, S, (1.0 / S) if ortho else 1.0,
              _stream())
        return dd
    dd = _dft_planes_adj(dY, S, ortho)
    return dd if dd.shape[-1] == S else dd[:, :S, :S].contiguous()


def _dft_planes_adj(dY, S, ortho):
    """adjoint of _dft_planes_fwd: dY [P, 2S, Whp] -> [P, Sp, Sp] with the S x S result in [:S, :S] (zeros around)"""
    P = dY.shape[0]
    fw_cos, fw_sin, fh = _dft_mats(S, dY.device, ortho)
    Whp, Sp = fw_cos.shape
    dt_re = empty((P * Sp, Whp), dY)
    dt_im = empty((P * Sp, Whp), dY)
    # dTre_p = fh[:, :Sp]^T @ dY_p ;  dTim_p = fh[:, Sp:]^T @ dY_p        (A[k][m] = fh[k][m(+Sp)]; rows m >= S come out zero)
    _gemm(fh, dY, dt_re, Sp, Whp, 2 * S, 2 * Sp, Whp, Whp, 1, 1, 0, batch=P, strideA=0, strideB=2 * S * Whp,
          strideC=Sp * Whp)
    _gemm(fh, dY, dt_im, Sp, Whp, 2 * S, 2 * Sp, Whp, Whp, 1, 1, 0, batch=P, strideA=0, strideB=2 * S * Whp,
          strideC=Sp * Whp, a_off=Sp)
    dd = gemm_nn(dt_re, fw_cos)                     # [P*Sp, Whp] @ [Whp, Sp]
    gemm_nn(dt_im, fw_sin, out=dd, accumulate=True)
    return dd.view(P, Sp, Sp)


# ---------------------------------------------------------------------------------------------
# ResNet-variant helpers (pool.hip)
# ---------------------------------------------------------------------------------------------
def avgpool_fwd(x, k):
    _chk(x)
    N, H, W, Cc = x.shape
    assert H % k == 0 and W % k == 0
    y = empty((N, H // k, W // k, Cc), x)
    _call("ud_avgpool_fwd", _p(x), _p(y), N, H // k, W // k, Cc, k, _stream())
    return y


def avgpool_bwd(dy, k):
    _chk(dy)
    N, Ho, Wo, Cc = dy.shape
    dx = empty((N, Ho * k, Wo * k, Cc), dy)
    _call("ud_avgpool_bwd", _p(dy), _p(dx), N, Ho, Wo, Cc, k, _stream())
    return dx


def adaptive_avgpool_fwd(x, Ho, Wo):
    """F.adaptive_avg_pool2d on pixel-major x [N, H, W, C] to (Ho, Wo), any Ho <= H, Wo <= W (ATen's window rule)."""
    _chk(x)
    N, H, W, Cc = x.shape
    y = empty((N, Ho, Wo, Cc), x)
    _call("ud_adaptive_avgpool_fwd", _p(x), _p(y), N, H, W, Ho, Wo, Cc, _stream())
    return y


def adaptive_avgpool_bwd(dy, H, W):
    _chk(dy)
    N, Ho, Wo, Cc = dy.shape
    dx = empty((N, H, W, Cc), dy)
    _call("ud_adaptive_avgpool_bwd", _p(dy), _p(dx), N, H, W, Ho, Wo, Cc, _stream())
    return dx


def maxpool3s2_fwd(x):
    _chk(x)
    N, H, W, Cc = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = empty((N, Ho, Wo, Cc), x)
    arg = torch.empty((N, Ho, Wo, Cc), dtype=torch.uint8, device=x.device)
    _call("ud_maxpool3s2_fwd", _p(x), _p(y), C.c_void_p(arg.data_ptr()), N, H, W, Cc, _stream())
    return y, arg


def maxpool3s2_bwd(dy, arg, H, W):
    _chk(dy)
    N, _, _, Cc = dy.shape
    dx = empty((N, H, W, Cc), dy)
    _call("ud_maxpool3s2_bwd", _p(dy), C.c_void_p(arg.data_ptr()), _p(dx), N, H, W, Cc, _stream())
    return dx


def add_act(a, b, act):
    _chk(a, b)
    y = torch.empty_like(a)
    _call("ud_add_act_fwd", _p(a), _p(b), int(act), _p(y), a.numel(), _stream())
    return y


def relu_bwd(dy, y):
    _chk(dy, y)
    g = torch.empty_like(dy)
    _call("ud_relu_bwd", _p(dy), _p(y), _p(g), dy.numel(), _stream())
    return g


def concat_channels(parts):
    """torch.cat(parts, dim=-1) for pixel-major tensors with equal leading dims."""
    _chk(*parts)
    lead = parts[0].shape[:-1]
    Cw = sum(p.shape[-1] for p in parts)
    M = parts[0].numel() // parts[0].shape[-1]
    out = empty((*lead, Cw), parts[0])
    off = 0
    for p in parts:
        _call("ud_copy_cols", _p(p), _p(out), M, p.shape[-1], Cw, off, 0, _stream())
        off += p.shape[-1]
    return out


def slice_channels(wide, off, Cn):
    _chk(wide)
    lead = wide.shape[:-1]
    M = wide.numel() // wide.shape[-1]
    out = empty((*lead, Cn), wide)
    _call("ud_copy_cols", _p(out), _p(wide), M, Cn, wide.shape[-1], off, 1, _stream())
    return out


# ---------------------------------------------------------------------------------------------
# losses on [N, d] feature vectors
# ---------------------------------------------------------------------------------------------
def aw_triplet(feat, n_real):
    """Asymmetrical weighted triplet loss (loss/triplet_loss.py:16-82): returns (loss[], dloss/dfeat[N,D])."""
    _chk(feat)
    N, D = feat.shape
    loss = empty((), feat)
    dfeat = torch.empty_like(feat)
    ws = empty((n_real * (N + 1),), feat)
    _call("ud_aw_triplet", _p(feat), N, D, int(n_real), _p(loss), _p(dfeat), _p(ws), _stream())
    return loss, dfeat


def loss_tail(cls_out, tgt, n_real, n_fake, feats, fm, sm, spatial, freq, weights):
    """The scalar tail of a pass's loss in two launches (ud_loss_tail_run): returns (vals[9], grads) with vals[0] the weighted
    total (see include/unidefense_hip.h) and grads a dict of d total / d input views into ONE flat buffer `grads["_flat"]` (so
    that the incoming loss gradient multiplies them all with one launch).  weights: (w_cls, w_fm, w_sm, w_trip, w_rec, w_freq)."""
    from .lib import LossTail
    _chk(cls_out, fm, sm, spatial, freq, *feats)
    N, Cc = cls_out.shape
    parts = [("cls", cls_out)] + [(f"feat{i}", f) for i, f in enumerate(feats)] + [("fm", fm), ("sm", sm), ("spatial", spatial),
                                                                                    ("freq", freq)]
    sizes = [(k, t) for k, t in parts if t is not None]
    total = sum((t.numel() + 3) // 4 * 4 for _, t in sizes)
    flat = empty((total,), cls_out)
    grads, off = {"_flat": flat}, 0
    for k, t in sizes:
        grads[k] = flat[off:off + t.numel()].view(t.shape)
        off += (t.numel() + 3) // 4 * 4
    vals = empty((9,), cls_out)
    t = LossTail()
    t.nfeat = len(feats)
    for i, f in enumerate(feats):
        assert f.shape[0] == N
        t.feat[i], t.dfeat[i], t.D[i] = f.data_ptr(), grads[f"feat{i}"].data_ptr(), f.shape[1]
    t.cls, t.tgt, t.dcls = cls_out.data_ptr(), tgt.data_ptr(), grads["cls"].data_ptr()
    t.N, t.C, t.R, t.F = N, Cc, int(n_real), int(n_fake)
    if fm is not None:
        t.fm, t.dfm, t.nfm = fm.data_ptr(), grads["fm"].data_ptr(), fm.numel()
    if sm is not None:
        t.sm, t.dsm, t.nsm = sm.data_ptr(), grads["sm"].data_ptr(), sm.numel()
    if spatial is not None:
        t.spatial, t.dspatial = spatial.data_ptr(), grads["spatial"].data_ptr()
    if freq is not None:
        t.freq, t.dfreq = freq.data_ptr(), grads["freq"].data_ptr()
    t.w_cls, t.w_fm, t.w_sm, t.w_trip, t.w_rec, t.w_freq = [float(w) for w in weights]
    t.vals = vals.data_ptr()
    ws = None
    if feats:
        ws = empty((_call("ud_loss_tail_ws_floats", N, int(n_real), len(feats)),), cls_out)
        t.ws = ws.data_ptr()
    _call("ud_loss_tail_run", C.byref(t), _stream())
    return vals, grads


# ---------------------------------------------------------------------------------------------
# pass-2 input perturbations on NCHW planes (perturb.hip; model/unidefense.py:177-198 of the reference)
# ---------------------------------------------------------------------------------------------
_PERTURB_WS = {}


def gather2d(x, iy, ix):
    """x [.., H, W] -> out[.., y, x] = x[.., iy[y], ix[x]]  (iy, ix: int32 device vectors)."""
    _chk(x)
    H, W = x.shape[-2:]
    out = torch.empty_like(x)
    _call("ud_gather2d", _p(x), _p(out), _p(iy), _p(ix), x.numel() // (H * W), H, W, _stream())
    return out


def blur5_reflect(x, taps):
    _chk(x)
    H, W = x.shape[-2:]
    out = torch.empty_like(x)
    _call("ud_blur5_reflect", _p(x), _p(out), x.numel() // (H * W), H, W, float(taps[0]), float(taps[1]),
          float(taps[2]), _stream())
    return out


def amp_mix(Ya, Yb, lmda, S, planes_per_sample):
    """Spectra [P, 2S, Whp] (dft_rfft2_planes layout) -> w * (l|A| + (1-l)|B|) * exp(i angle(A))."""
    _chk(Ya, Yb, lmda)
    P, S2, Whp = Ya.shape
    assert S2 == 2 * S and Yb.shape == Ya.shape and lmda.numel() * planes_per_sample == P
    out = torch.empty_like(Ya)
    _call("ud_amp_mix", _p(Ya), _p(Yb), _p(lmda), _p(out), P, S, Whp, planes_per_sample, _stream())
    return out


def efdm(content, style, lmda, rows_per_sample):
    """content/style [rows, L]; lmda [rows / rows_per_sample] -> rank-matched mix (model/modules.py:58-76)."""
    _chk(content, style, lmda)
    rows, L = content.shape
    assert style.shape == content.shape and lmda.numel() * rows_per_sample == rows
    need = _call("ud_efdm_ws_bytes", rows, L)
    key = content.device.index
    ws = _PERTURB_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = _PERTURB_WS[key] = torch.empty(need, dtype=torch.uint8, device=content.device)
    out = torch.empty_like(content)
    _call("ud_efdm", _p(content), _p(style), _p(lmda), _p(out), rows, L, rows_per_sample, _p(ws), ws.numel(), _stream())
    return out


def coral_moments(x, chunks=16):
    """x [N, 3, H, W] -> fp64 [N, 9]: sum x_c (3), sum x_c x_d (00 01 02 11 12 22)."""
    _chk(x)
    N = x.shape[0]
    HW = x.shape[-1] * x.shape[-2]
    part = torch.empty((N, chunks, 9), dtype=torch.float64, device=x.device)
    _call("ud_coral_moments", _p(x), _p(part), N, HW, chunks, _stream())
    return part.sum(1)


def affine3(x, M):
    """out[n, c] = sum_k M[n, c, k] x[n, k] + M[n, c, 3]  on [N, 3, H, W]."""
    _chk(x, M)
    assert M.shape == (x.shape[0], 3, 4)
    out = torch.empty_like(x)
    _call("ud_affine3", _p(x), _p(M), _p(out), x.shape[0], x.shape[-1] * x.shape[-2], _stream())
    return out


# ---------------------------------------------------------------------------------------------
# fused MBConv path: deferred BatchNorm (csrc/fused.hip; include/unidefense_hip.h "deferred normalisation")
# ---------------------------------------------------------------------------------------------
from .lib import BnRef      # noqa: E402


class DeferredBN:
    """A training-mode BatchNorm that is never applied as a pass of its own: `acc` holds [sum | sumsq] (2C doubles,
    zero until the producer's kernels add into it), consumers apply act(gamma (x - mean) invstd + beta) on load."""

    def __init__(self, acc, C, count, gamma, beta, eps, act, momentum=0.0, running_mean=None, running_var=None):
        self.acc, self.C, self.count = acc, C, float(count)
        self.gamma, self.beta, self.eps, self.act = gamma, beta, float(eps), int(act)
        self.momentum, self.running_mean, self.running_var = float(momentum), running_mean, running_var
        self._update_pending = running_mean is not None

    def ref(self, update=False):
        """ctypes struct for a kernel call.  update=True hands the running statistics to THIS call (the kernel moves
        them once); every BatchNorm's first consumer does so, later ones must not."""
        r = BnRef()
        r.sum, r.sumsq = self.acc.data_ptr(), self.acc.data_ptr() + 8 * self.C
        r.gamma, r.beta = self.gamma.data_ptr(), self.beta.data_ptr()
        r.inv_count = 1.0 / self.count
        r.unbias = self.count / (self.count - 1.0) if self.count > 1 else 1.0
        r.eps, r.momentum, r.act, r.G = self.eps, self.momentum, self.act, 1
        if update and self._update_pending:
            r.running_mean, r.running_var = self.running_mean.data_ptr(), self.running_var.data_ptr()
            self._update_pending = False
        else:
            r.running_mean = r.running_var = None
        return r


def _pd(t, off_doubles=0):
    return C.c_void_p(t.data_ptr() + 8 * off_doubles)


def _fused_ws(ref, G, R, C_, per_group, min_rows=8):
    """Scratch pointer for the two-launch form of a fused reduction (None when it runs as one launch of atomics)."""
    need = _call("ud_fused_reduce_ws_doubles", G, R, C_, int(per_group), min_rows)
    return _ws64(ref, need)


def _ws64_t(ref, need):
    """the (device, branch) fp64 scratch as a tensor of at least `need` doubles"""
    ws = _REDUCE_WS.get(_key(ref))
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, _REDUCE_WS_MIN), dtype=torch.float64, device=ref.device)
        _REDUCE_WS[_key(ref)] = ws
    return ws


def _ws64(ref, need):
    if need <= 0:
        return None
    ws = _REDUCE_WS.get(_key(ref))
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, _REDUCE_WS_MIN), dtype=torch.float64, device=ref.device)
        _REDUCE_WS[_key(ref)] = ws
    return C.c_void_p(ws.data_ptr())


def colstats(x2, acc, G=1, R=None):
    """acc[0:G*C] += column sums of x2 [G*R, C], acc[G*C:2*G*C] += column sums of squares."""
    h = _act(x2)
    Cc = x2.shape[-1]
    R = x2.numel() // (G * Cc) if R is None else R
    _call("ud_colstats", _p(x2), G, R, Cc, _pd(acc), _pd(acc, G * Cc), _fused_ws(x2, G, R, Cc, G != 1), h, _stream())


def colsum_bn(x, bn, G, R, out, update=False):
    h = _act(x)
    Cc = x.shape[-1]
    _call("ud_colsum_bn", _p(x), C.byref(bn.ref(update)), G, R, Cc, _pd(out), _fused_ws(x, G, R, Cc, True), h, _stream())


def colsum_bn_amax(x, bn, G, R, out, update=False):
    """colsum_bn (fp32) + the 256 slots holding max |act(bn(x))| (returned): the scale se_scale_bn_planes writes its planes with"""
    _chk(x)
    Cc = x.shape[-1]
    amax = zeros((256,), x)
    _call("ud_colsum_bn_amax", _p(x), C.byref(bn.ref(update)), G, R, Cc, _pd(out), _fused_ws(x, G, R, Cc, True), _p(amax), _stream())
    return amax


def se_scale_bn_planes(x, bn, s, G, R, amax):
    """act(bn(x)) * sigmoid(s) written straight into prec-2 Planes over [G R] x C (the project conv's GEMM operand)"""
    _chk(x, s)
    Cc = x.shape[-1]
    pl = Planes(G * R, Cc, x, 2, False)
    _call("ud_se_scale_bn_planes", _p(x), C.byref(bn.ref()), _p(s), _p(pl.buf), pl.panel, pl.plane, _p(pl.inv), _p(amax), G, R, Cc,
          _stream())
    return pl


def se_scale_bn_plane_half(x, bn, s, G, R):
    """half storage: act(bn(x)) * sigmoid(s) laid straight into the prec-1 plane over [G R] x C (the project conv's GEMM operand)"""
    _act(x)
    _chk(s)
    Cc = x.shape[-1]
    pl = Planes(G * R, Cc, x, 1, False)
    _call("ud_se_scale_bn_plane_half", _p(x), C.byref(bn.ref()), _p(s), _p(pl.buf), pl.panel, _p(pl.inv), G, R, Cc, _stream())
    return pl


def coldot_bn(dy, x, bn, G, R, out):
    h = _act(dy, x)
    Cc = x.shape[-1]
    _call("ud_coldot_bn", _p(dy), _p(x), C.byref(bn.ref()), G, R, Cc, _pd(out), _fused_ws(x, G, R, Cc, True), h,
          _stream())


def fc_fwd_d(xsum, xscale, W, b, N):
    _chk(W, b)
    O, I = W.shape
    y = empty((N, O), W)
    _call("ud_fc_fwd_d", _pd(xsum), float(xscale), _p(W), _p(b), _p(y), N, I, O, _stream())
    return y


def bn_apply(x, bn, G, R, update=False):
    """y = act(bn(x)), materialised."""
    h = _act(x)
    y = torch.empty_like(x)
    _call("ud_bn_apply", _p(x), C.byref(bn.ref(update)), _p(y), G, R, x.shape[-1], h, _stream())
    return y


def se_scale_bn(x, bn, s, G, R, want_absmax=False):
    h = _act(x)
    _chk(s)
    y = torch.empty_like(x)
    amax = amax_slots(x, want_absmax)
    _call("ud_se_scale_bn", _p(x), C.byref(bn.ref()), _p(s), _p(y), G, R, x.shape[-1], h, _p(amax), _stream())
    y._ud_absmax = amax
    return y


_RESIDUAL_PLANES = True          # A/B: tools/run_with.py kernels._RESIDUAL_PLANES=False


def residual_bn(x, bn, keep, inv_keep, skip, G, R, update=False, want_absmax=False, planes_for=None):
    """out = bn(x) [* keep / keep_prob] [+ skip].  planes_for = (M, N, w2) of the 1x1 conv that reads `out` next (the following
    block's expand conv): where that conv runs on the planes GEMM, `out` is ALSO written as its fp16 x 2 planes
    (out._ud_planes; scale from an a-priori bound, ud_residual_bn_planes) — no split pass over the block output."""
    h = _act(x, skip)
    _chk(keep)
    out = torch.empty_like(x)
    amax = amax_slots(x, want_absmax)
    Cc = x.shape[-1]
    skip_amax = getattr(skip, "_ud_absmax", None) if skip is not None else None
    if (planes_for is not None and _RESIDUAL_PLANES and _RFFT_PLANES and h == 0 and Cc % 4 == 0 and CFG.spectral_p2 != "off"
            and (skip is None or skip_amax is not None)
            and spectral_takes_planes(planes_for[0], planes_for[1], Cc, planes_for[2], want_stats=True)):
        pl = Planes(G * R, Cc, x, 2, False)
        _call("ud_residual_bn_planes", _p(x), C.byref(bn.ref(update)), _p(keep), float(inv_keep), _p(skip), _p(skip_amax), _p(out),
              _p(pl.buf), pl.panel, pl.plane, _p(pl.inv), G, R, Cc, _p(amax), _stream())
        out._ud_planes = pl
    else:
        _call("ud_residual_bn", _p(x), C.byref(bn.ref(update)), _p(keep), float(inv_keep), _p(skip), _p(out), G, R,
              Cc, h, _p(amax), _stream())
    out._ud_absmax = amax
    return out


def normbwd_sums(x, dy, keep, inv_keep, bn, dy_is_dz, G, R, sacc):
    """sacc[0:C] += sum dz, sacc[C:2C] += sum dz * xhat; a 3C accumulator also receives sacc[2C:3C] += sum dz^2 (rounded up: the
    energy normbwd_apply_planes bounds its result by)."""
    h = _act(x, dy)
    _chk(keep)
    Cc = x.shape[-1]
    _call("ud_normbwd_sums", _p(x), _p(dy), _p(keep), float(inv_keep), C.byref(bn.ref()), int(dy_is_dz), G, R, Cc,
          _pd(sacc), _pd(sacc, Cc), _pd(sacc, 2 * Cc) if sacc.numel() >= 3 * Cc else None, _fused_ws(x, G, R, Cc, False), h,
          _stream())


_NORMBWD_PLANES = True          # A/B: tools/run_with.py kernels._NORMBWD_PLANES=False


def normbwd_planes_ok(x, ctx):
    """does the BatchNorm backward in front of a 1x1 conv write its result as that conv's GEMM planes itself?  (the conv's
    backward runs on the planes GEMM — ctx.plans — and the tensor has whole channel quads.)  1: fp32 — prec-2 planes, scaled by the
    energy bound (a 3C sum accumulator); 2: half storage — the half result laid into the prec-1 plane as it is; 0: no."""
    if not (_NORMBWD_PLANES and _RFFT_PLANES and ctx.plans is not None and x.shape[-1] % 4 == 0 and CFG.spectral_p2 != "off"):
        return 0
    return 1 if x.dtype == torch.float32 else 2 if (_P1_DIRECT and x.dtype == torch.float16 and x.shape[-1] % 8 == 0) else 0


def normbwd_apply_planes(x, dy, keep, inv_keep, bn, dy_is_dz, G, R, sacc, sacc_local=None):
    """normbwd_apply whose result is prec-2 Planes over [G R] x C (no fp32 tensor, no split pass); sacc: 3C sums over all ranks
    (sum dz | sum dz xhat | energy sum dz^2).  Returns (Planes, dgamma, dbeta)."""
    Cc = x.shape[-1]
    loc = sacc if sacc_local is None else sacc_local
    if x.dtype == torch.float16:          # the mixed-precision mode: one plane, values unchanged
        _act(x, dy)
        _chk(keep)
        pl = Planes(G * R, Cc, x, 1, False)
        dg = empty((Cc,), x)
        db = empty((Cc,), x)
        _call("ud_normbwd_apply_plane_half", _p(x), _p(dy), _p(keep), float(inv_keep), C.byref(bn.ref()), int(dy_is_dz), _pd(sacc),
              _pd(sacc, Cc), _pd(loc), _pd(loc, Cc), G, R, Cc, _p(pl.buf), pl.panel, _p(pl.inv), _p(dg), _p(db), _stream())
        return pl, dg, db
    _chk(x, dy, keep)
    assert x.dtype == torch.float32 and dy.dtype == torch.float32
    assert sacc.numel() >= 3 * Cc
    pl = Planes(G * R, Cc, x, 2, False)
    dg = empty((Cc,), x)
    db = empty((Cc,), x)
    _call("ud_normbwd_apply_planes", _p(x), _p(dy), _p(keep), float(inv_keep), C.byref(bn.ref()), int(dy_is_dz), _pd(sacc),
          _pd(sacc, Cc), _pd(loc), _pd(loc, Cc), _pd(sacc, 2 * Cc), G, R, Cc, _p(pl.buf), pl.panel, pl.plane, _p(pl.inv), _p(dg),
          _p(db), _stream())
    return pl, dg, db


def normbwd_apply(x, dy, keep, inv_keep, bn, dy_is_dz, G, R, sacc, sacc_local=None, want_dbeta=True, want_absmax=False):
    """Returns (dx, dgamma, dbeta).  sacc: sums over all ranks; sacc_local: this rank's (default: the same)."""
    h = _act(x, dy)
    _chk(keep)
    Cc = x.shape[-1]
    loc = sacc if sacc_local is None else sacc_local
    dx = torch.empty_like(x)
    dg = empty((Cc,), x)
    db = empty((Cc,), x) if want_dbeta else None
    amax = amax_slots(x, want_absmax)
    _call("ud_normbwd_apply", _p(x), _p(dy), _p(keep), float(inv_keep), C.byref(bn.ref()), int(dy_is_dz), _pd(sacc),
          _pd(sacc, Cc), _pd(loc), _pd(loc, Cc), G, R, Cc, _p(dx), _p(dg), _p(db), h, _p(amax), _stream())
    dx._ud_absmax = amax
    return dx, dg, db


def expand_bwd_fused_ok(e, w2, dy_is_dz):
    """does ud_pw_bwd_fused take this expand conv's backward (fp32 storage, a thin (Ce, Cin) pair it is built for)?"""
    return (CFG.expand_bwd_fused and e.dtype == torch.float32 and bool(dy_is_dz) and w2.dtype == torch.float32 and
            _lib.call("ud_pw_bwd_fused_ok", int(w2.shape[0]), int(w2.shape[1])) == 1)


def expand_bwd_fused(e, dz, bn, sacc, sacc_local, x2, w2, add=None):
    """BatchNorm-0 backward + the expand conv's weight and data gradient in one pass over (dz, e): returns (dx, dw, dgamma, dbeta).
    e, dz: [.., Ce] (M rows); x2 [M, Cin]; w2 [Ce, Cin]; add: a term of dx to add onto IN PLACE (dx is add then), or None."""
    _chk(e, dz, x2, w2)
    Ce, Cin = w2.shape
    M = x2.shape[0]
    assert e.numel() == M * Ce and dz.numel() == M * Ce and x2.is_contiguous() and w2.is_contiguous()
    loc = sacc if sacc_local is None else sacc_local
    dx = add if add is not None else torch.empty_like(x2)
    if add is not None:
        _chk(add)
        assert add.numel() == M * Cin
    dw = empty((Ce, Cin), x2)
    dg = empty((Ce,), x2)
    db = empty((Ce,), x2)
    grid = _lib.call("ud_pw_bwd_fused_grid", M)
    part = empty((grid, Ce * Cin), x2)
    _call("ud_pw_bwd_fused", _p(e), _p(dz), C.byref(bn.ref()), _pd(sacc), _pd(sacc, Ce), _pd(loc), _pd(loc, Ce), _p(x2), _p(w2),
          _p(add), M, Ce, Cin, _p(dx), _p(dw), _p(part), _p(dg), _p(db), _stream())
    return dx, dw, dg, db


def project_bwd_fused_ok(d, w2, HW):
    """do ud_pj_bwd_fused_a / _b take this project conv's backward (fp32 storage, a thin (Ce, Co) pair they are built for, whole
    32-row tiles per sample)?"""
    return (CFG.project_bwd_fused and d.dtype == torch.float32 and w2.dtype == torch.float32 and
            (CFG.project_bwd_fused_wide or w2.shape[0] != 56) and (CFG.project_fused_narrow or w2.shape[0] != 24) and
            _lib.call("ud_pj_bwd_fused_ok", int(w2.shape[1]), int(w2.shape[0]), int(HW)) == 1)


def project_fwd_fused_ok(d, w2, HW):
    """does ud_pj_fwd_fused take this project conv's forward (and ud_pj_bwd_fused_a / _b its backward)?"""
    return (project_bwd_fused_ok(d, w2, HW) and _lib.call("ud_pj_fwd_fused_ok", int(w2.shape[1]), int(w2.shape[0]), int(HW)) == 1)


def project_fwd_fused(d, bn, s, w2, N, HW, stats=None):
    """p[N HW, Co] = (act(bn(d)) sigmoid(s)) w2^T in one pass over d (the gated tensor is not written); stats: fp64 [2 Co] accumulator
    that receives p's column sums and sums of squares.  Returns (p, ctx): the context of a backward that ud_pj_bwd_fused_a / _b run."""
    _chk(d, w2, s)
    Co, Ce = w2.shape
    p = empty((N * HW, Co), d)
    _call("ud_pj_fwd_fused", _p(d), C.byref(bn.ref()), _p(s), _p(w2), N, HW, Ce, Co, _p(p), _pd(stats) if stats is not None else None,
          _pd(stats, Co) if stats is not None else None, _stream())
    ctx = SpectralCtx()
    ctx.plans, ctx.x, ctx.w, ctx.dy, ctx.M, ctx.N, ctx.K = None, None, w2, None, N * HW, Co, Ce
    return p, ctx


def project_bwd_fused_a(d, bn, s, dp2, w2, N, HW, dgate):
    """dw[Co, Ce] = dp^T (act(bn(d)) sigmoid(s)) and dgate[N, Ce] (fp64, zeroed by the caller) += sum_hw (dp w) act(bn(d)) in one
    pass over d [N, .., Ce]; dp2 [N HW, Co]; w2 [Co, Ce]; s [N, Ce]."""
    _chk(d, dp2, w2, s)
    Co, Ce = w2.shape
    assert d.numel() == N * HW * Ce and dp2.numel() == N * HW * Co and s.numel() == N * Ce and dp2.is_contiguous()
    dw = empty((Co, Ce), d)
    grid = _lib.call("ud_pj_bwd_fused_grid", N, HW, Ce, Co)
    part = empty((grid, Co * Ce), d)          # (an upper bound: a workgroup's partial covers its chunk of the Ce columns)
    _call("ud_pj_bwd_fused_a", _p(d), _p(dp2), C.byref(bn.ref()), _p(s), _p(w2), N, HW, Ce, Co, _p(dw), _pd(dgate), _p(part), _stream())
    return dw


def project_bwd_fused_b(d, bn, s, dpool, inv_hw, dp2, w2, N, HW, sacc):
    """dz = ((dp w) sigmoid(s) + dpool inv_hw) act'(bn(d)) [like d]; sacc (fp64 [2 Ce]) += sum dz, sum dz xhat"""
    _chk(d, dp2, w2, s, dpool)
    Co, Ce = w2.shape
    dz = torch.empty_like(d)
    _call("ud_pj_bwd_fused_b", _p(d), _p(dp2), C.byref(bn.ref()), _p(s), _p(dpool), float(inv_hw), _p(w2), N, HW, Ce, Co, _p(dz),
          _pd(sacc), _pd(sacc, Ce), _stream())
    return dz


def normbwd_apply_mix(x, dz, bn, G, R, sacc, diff, dalpha_acc, sacc_local=None, energy=None):
    """diff = freq - spat (irfft2_mix).  energy: C zeroed doubles that receive sum_rows dd^2 per channel (rfft2_ex_planes' bound)."""
    h = _act(x, dz, diff)
    Cc = x.shape[-1]
    loc = sacc if sacc_local is None else sacc_local
    dd = torch.empty_like(x)
    dg = empty((Cc,), x)
    db = empty((Cc,), x)
    _call("ud_normbwd_apply_mix", _p(x), _p(dz), C.byref(bn.ref()), _pd(sacc), _pd(sacc, Cc), _pd(loc), _pd(loc, Cc),
          _p(diff), G, R, Cc, _p(dd), _pd(dalpha_acc), _p(dg), _p(db), _pd(energy) if energy is not None else None, h, _stream())
    return dd, dg, db


def gate_grad_from_acc(acc, alpha):
    out = empty((), alpha)
    _call("ud_gate_grad_from_acc", _pd(acc), _p(alpha), _p(out), _stream())
    return out


def se_bwd(dgate, s2, s1, We, Wr, pool, pool_scale):
    """Backward of the two SE FCs (model.py:119-121) in two launches.  dgate: fp64 [N, C] (gradient of the gate
    sigmoid(s2)); pool: fp64 pooled SUMS.  Returns (dpool, dWe, dbe, dWr, dbr)."""
    _chk(s2, s1, We, Wr)
    N, Cc = s2.shape
    Cs = s1.shape[1]
    ds1_acc = zeros64(N * Cs, s2)
    dWe = empty((Cc, Cs), s2)
    dbe = empty((Cc,), s2)
    _call("ud_se_bwd_a", _pd(dgate), _p(s2), _p(s1), _p(We), _pd(ds1_acc), _p(dWe), _p(dbe), N, Cc, Cs, _stream())
    dpool = empty((N, Cc), s2)
    dWr = empty((Cs, Cc), s2)
    dbr = empty((Cs,), s2)
    _call("ud_se_bwd_b", _pd(ds1_acc), _p(s1), _p(Wr), _pd(pool), float(pool_scale), _p(dpool), _p(dWr), _p(dbr), N, Cc,
          Cs, _stream())
    return dpool, dWe, dbe, dWr, dbr


def se_scale_bwd_bn(dc, x, bn, s, dpool, inv_hw, G, R, sacc):
    h = _act(dc, x)
    _chk(s, dpool)
    Cc = x.shape[-1]
    dz = torch.empty_like(x)
    _call("ud_se_scale_bwd_bn", _p(dc), _p(x), C.byref(bn.ref()), _p(s), _p(dpool), float(inv_hw), _p(dz), _pd(sacc),
          _pd(sacc, Cc), _fused_ws(x, G, R, Cc, False), G, R, Cc, h, _stream())
    return dz


def dwconv_bwd_data_bn(dy, gate_alpha, gate_mode, wt, add, x, bn, K, stride, pad_t, pad_l, sacc):
    """dz = (gate * dwconv_bwd_data(dy) + add) * act'(bn(x)); sacc += BatchNorm backward sums.  x: the conv's input."""
    h = _act(dy, add, x)
    _chk(wt)
    N, H, W, Cc = x.shape
    _, Ho, Wo, _ = dy.shape
    dz = torch.empty_like(x)
    ws = _ws64(x, _call("ud_dwconv_bwd_data_bn_ws_doubles", N, H, W, Cc, stride))
    _call("ud_dwconv_bwd_data_bn", _p(dy), _p(gate_alpha), int(gate_mode), _p(wt), _p(add), _p(x), C.byref(bn.ref()),
          _p(dz), _pd(sacc), _pd(sacc, Cc), ws, N, H, W, Cc, Ho, Wo, K, stride, pad_t, pad_l, h, _stream())
    return dz


def dwconv_bwd_data_ex(dy, gate_alpha, gate_mode, wt, add, K, stride, pad_t, pad_l, H, W):
    h = _act(dy, add)
    _chk(wt)
    N, Ho, Wo, Cc = dy.shape
    dx = empty((N, H, W, Cc), dy, dy.dtype)
    _call("ud_dwconv_bwd_data_ex", _p(dy), _p(gate_alpha), int(gate_mode), _p(wt), _p(add), _p(dx), N, H, W, Cc, Ho, Wo,
          K, stride, pad_t, pad_l, h, _stream())
    return dx


def dwconv_bwd_weight_ex(x, dy, gate_alpha, gate_mode, K, stride, pad_t, pad_l):
    h = _act(x, dy)
    N, H, W, Cc = x.shape
    _, Ho, Wo, _ = dy.shape
    lanes = Cc // 2 if h else Cc          # half: a lane owns two adjacent channels
    chunks = max(1, min(N * Ho, -(-_DW_WGRAD_THREADS // lanes)))
    part = empty((chunks, K * K, Cc), x)
    dwt = empty((Cc, K * K), x)
    _call("ud_dwconv_bwd_weight_ex", _p(x), _p(dy), _p(gate_alpha), int(gate_mode), _p(dwt), _p(part), chunks, N, H, W,
          Cc, Ho, Wo, K, stride, pad_t, pad_l, h, _stream())
    return dwt


def rfft2_ex(x, scale, w_interior=1.0, bn=None, want_act=False, gate_alpha=None, gate_mode=0, update=False,
             gate_acc=None, want_absmax=False):
    """rfft2 of act(bn(x)) (bn optional) [* gate].  Returns (Y, activated input or None[, gate gradient when gate_acc:
    sigmoid'(alpha) * sum of the 64 accumulator slots])."""
    h = _act(x)
    N, S, S2, Cc = x.shape
    assert S == S2
    Y = empty((N, S, S // 2 + 1, 2 * Cc), x, x.dtype)
    act = torch.empty_like(x) if (want_act and bn is not None) else None
    ggrad = empty((), x) if gate_acc is not None else None
    amax = amax_slots(x, want_absmax)
    Y._ud_absmax = amax
    tail = (C.byref(bn.ref(update)) if bn is not None else None, _p(act), _p(gate_alpha), int(gate_mode),
            _pd(gate_acc) if gate_acc is not None else None, _p(ggrad), h, _p(amax), _stream())
    if _fft_two_pass("rfft_ex", S, h):
        _call("ud_rfft2_two_pass", _p(x), _p(Y), _p(_fft_ws(N, S, Cc, x)), N, S, Cc, float(scale), float(w_interior), *tail)
    else:
        _call("ud_rfft2_ex", _p(x), _p(Y), N, S, Cc, float(scale), float(w_interior), *tail)
    return (Y, act, ggrad) if gate_acc is not None else (Y, act)


_RFFT_PLANES = True          # A/B: tools/run_with.py kernels._RFFT_PLANES=False


def rfft2_planes_ok(x, bn, stride_ok=True):
    """can ud_rfft2_ex_planes write this transform's result as the spectral GEMM's planes?  (fp32 storage, the one-kernel
    transform sizes, whole 32-column panels, and a conv shape the planes GEMM takes at all: spectral_takes_planes)"""
    if not (_RFFT_PLANES and x.dtype == torch.float32 and CFG.spectral_p2 != "off"):
        return False
    N, S, _, Cc = x.shape
    return S in (8, 16, 32, 12, 24, 48) and (2 * Cc) % 32 == 0 and not _fft_two_pass("rfft_ex", S, 0)


_RFFT_DW = True          # A/B: tools/run_with.py kernels._RFFT_DW=False


def rfft2_dw_ok(S, k, stride, pad):
    """can rfft2_ex_planes compute the depthwise conv of the same plane too?  (the 256 x 256 trunk's power-of-two maps, stride 1,
    'same' pads)"""
    return _RFFT_DW and stride == 1 and S in (8, 16, 32) and k in (3, 5) and tuple(pad) == ((k - 1) // 2,) * 4


def rfft2_ex_planes(x, scale, w_interior=1.0, bn=None, want_act=False, gate_alpha=None, gate_mode=0, update=False,
                    gate_acc=None, energy=None, dw_wt=None, dw_k=0):
    """rfft2_ex whose result is written straight into fp16 x 2 planes (P32 layout over [N S (S/2+1)] x 2C, prec 2) with the scale
    of an a-priori BOUND of |Y| (csrc/fft.hip: PlanesOut) instead of fp32 + ud_split_planes_h2t.  bn given: the bound comes from
    count (gamma^2 + beta^2); else energy [C] fp64: a per-channel upper bound of sum_{n,h,w} x^2 (the kernel multiplies the bound by
    the gate factor it applies to the result).  dw_wt / dw_k: also the stride-1 depthwise conv of the activated plane (taps
    tap-major [k*k][C]); its result is appended.  Returns (Planes, activated input or None[, gate gradient][, conv result])."""
    _chk(x)
    N, S, S2, Cc = x.shape
    assert S == S2 and (bn is not None or energy is not None)
    R = N * S * (S // 2 + 1)
    pl = Planes(R, 2 * Cc, x, 2, False)
    act = torch.empty_like(x) if (want_act and bn is not None) else None
    ggrad = empty((), x) if gate_acc is not None else None
    pre = float(scale) * max(1.0, float(w_interior)) * S
    if energy is None:
        pre *= math.sqrt(float(bn.count))
    spat = torch.empty_like(x) if dw_k else None
    _call("ud_rfft2_ex_planes", _p(x), _p(pl.buf), pl.panel, pl.plane, _p(pl.inv), pre, _pd(energy) if energy is not None else None,
          N, S, Cc, float(scale), float(w_interior), C.byref(bn.ref(update)) if bn is not None else None, _p(act), _p(gate_alpha),
          int(gate_mode), _pd(gate_acc) if gate_acc is not None else None, _p(ggrad), _p(dw_wt), _p(spat), int(dw_k), _stream())
    out = (pl, act, ggrad) if gate_acc is not None else (pl, act)
    return out + (spat,) if dw_k else out


def irfft2_mix(Y, scale, spat, alpha, acc):
    """(y, diff) = SF mix of spat with freq = irfft2(Y), and freq - spat; acc += [sum y | sum y^2]."""
    h = _act(Y, spat)
    _chk(alpha)
    N, S, Wh, C2 = Y.shape
    Cc = C2 // 2
    assert spat.shape == (N, S, S, Cc)
    y = torch.empty_like(spat)
    fr = torch.empty_like(spat)
    if _fft_two_pass("irfft_mix", S, h):
        _call("ud_irfft2_two_pass", _p(Y), _p(y), _p(_fft_ws(N, S, Cc, Y)), N, S, Cc, float(scale), 1.0, _p(spat), _p(alpha),
              _p(fr), _pd(acc), _pd(acc, Cc), h, _stream())
    else:
        _call("ud_irfft2_mix", _p(Y), _p(y), N, S, Cc, float(scale), 1.0, _p(spat), _p(alpha), _p(fr), _pd(acc),
              _pd(acc, Cc), h, _stream())
    return y, fr


# ---------------------------------------------------------------------------------------------
# LDS-tiled depthwise conv, stride 1 (csrc/dwtile.hip): deferred BatchNorm applied while the halo tile is staged
# ---------------------------------------------------------------------------------------------
def _bnp(bn, update=False):
    return C.byref(bn.ref(update)) if bn is not None else None


def dwtile_fwd(x, wt, K, pad_t, pad_l, Ho, Wo, bn=None, stats=None, update=False, stride=1):
    """y = dwconv(act(bn(x))) (bn: DeferredBN or None), stride 1 / 2; stats: 2C zeroed doubles that receive sum y | sum y^2."""
    h = _act(x)
    _chk(wt)
    N, H, W, Cc = x.shape
    y = empty((N, Ho, Wo, Cc), x, x.dtype)
    ws = None
    if stats is not None:
        ws = _ws64(x, _call("ud_dwtile_ws_doubles", N, Ho, Wo, Cc))
    _call("ud_dwtile", _p(x), _bnp(bn, update), _p(wt), _p(y), N, H, W, Cc, Ho, Wo, K, pad_t, pad_l, 0, None, 0, None, None,
          None, 1 if stats is not None else 0, _pd(stats) if stats is not None else None,
          _pd(stats, Cc) if stats is not None else None, ws, int(stride), h, _stream())
    return y


def dwtile_bwd_data(dy, wt, K, pad_t, pad_l, H, W, gate_alpha=None, gate_mode=0, add=None, x=None, bn=None, sacc=None,
                    stride=1):
    """da = gate * dwconv_bwd_data(dy) [+ add]; with (x, bn, sacc): dz = da * act'(bn(x)), sacc += BatchNorm backward sums."""
    h = _act(dy, add, x)
    _chk(wt)
    N, Ho, Wo, Cc = dy.shape
    out = empty((N, H, W, Cc), dy, dy.dtype)
    ws = None
    if bn is not None:
        ws = _ws64(dy, _call("ud_dwtile_ws_doubles", N, H, W, Cc))
    _call("ud_dwtile", _p(dy), None, _p(wt), _p(out), N, Ho, Wo, Cc, H, W, K, K - 1 - pad_t, K - 1 - pad_l, 1,
          _p(gate_alpha), int(gate_mode), _p(add), _p(x) if bn is not None else None, _bnp(bn), 2,
          _pd(sacc) if bn is not None else None, _pd(sacc, Cc) if bn is not None else None, ws, int(stride), h, _stream())
    return out


_DWTILE_PART = {}


# The depthwise weight gradients come out of their kernels as partial rows [parts][K*K][C] that a small launch folds into the
# parameter's layout: 47 such launches per UDEB4 backward (~5 us each, the floor).  While a tape's backward collects them
# (begin_wgrad_folds ... flush_wgrad_folds: Tape.backward), every conv keeps its rows in a buffer of its own and ONE launch folds
# them all at the end (ud_dwtile_wgrad_finalize_multi).  A gradient that is read before the end — a second use of the parameter,
# a data-parallel reducer taking it as soon as it is complete — flushes first (Tape.add_param_grad).
_WGRAD_FOLDS = None
_WGRAD_FOLD_DEFER = True          # A/B: tools/run_with.py kernels._WGRAD_FOLD_DEFER=False


def begin_wgrad_folds():
    global _WGRAD_FOLDS
    if _WGRAD_FOLDS:
        flush_wgrad_folds()          # a backward started inside another one's: the outer tape's pending folds are done first, not dropped
    _WGRAD_FOLDS = [] if _WGRAD_FOLD_DEFER else None


def flush_wgrad_folds(end=False):
    """fold every collected weight gradient now (end: and stop collecting)"""
    global _WGRAD_FOLDS
    items = _WGRAD_FOLDS
    if items:
        from .lib import WgradFold
        arr = (WgradFold * len(items))()
        for a, (part, nparts, K_, Cc, alpha, mode, dwt) in zip(arr, items):
            a.part, a.dwt, a.gate_alpha = part.data_ptr(), dwt.data_ptr(), (alpha.data_ptr() if alpha is not None else None)
            a.nparts, a.K, a.C, a.gate_mode = int(nparts), int(K_), int(Cc), int(mode)
        _call("ud_dwtile_wgrad_finalize_multi", arr, len(items), _stream())
        for it in items:
            it[6]._ud_deferred = False
    _WGRAD_FOLDS = None if (end or items is None) else []


def _wgrad_part(like, need):
    """the partial-row buffer of one depthwise weight gradient: the shared scratch, or (folds being collected) its own"""
    if _WGRAD_FOLDS is not None:
        return torch.empty(need, dtype=torch.float32, device=like.device), True
    part = _DWTILE_PART.get(_key(like))
    if part is None or part.numel() < need:
        part = _DWTILE_PART[_key(like)] = torch.empty(need, dtype=torch.float32, device=like.device)
    return part, False


def _defer_wgrad_fold(part, nparts, K_, Cc, gate_alpha, gate_mode, dwt):
    dwt._ud_deferred = True
    _WGRAD_FOLDS.append((part, nparts, K_, Cc, gate_alpha if gate_mode else None, gate_mode, dwt))


def dwtile_bwd_weight(x, dy, K, pad_t, pad_l, bn=None, gate_alpha=None, gate_mode=0, stride=1):
    """dw[C, K*K] = gate * sum act(bn(x))(window) * dy, stride 1; x: the conv's RAW input when bn is given."""
    h = _act(x, dy)
    N, H, W, Cc = x.shape
    _, Ho, Wo, _ = dy.shape
    rows = _call("ud_dwtile_wgrad_part_rows", N, Ho, Wo)
    part, defer = _wgrad_part(x, rows * K * K * Cc)
    dwt = empty((Cc, K * K), x)
    nparts = _call("ud_dwtile_wgrad", _p(x), _bnp(bn), _p(dy), _p(gate_alpha), int(gate_mode), None if defer else _p(dwt), _p(part),
                   rows, N, H, W, Cc, Ho, Wo, K, pad_t, pad_l, int(stride), h, _stream())
    if defer:
        _defer_wgrad_fold(part, nparts, K, Cc, gate_alpha, gate_mode, dwt)
    return dwt


def dwtile_bwd(dy, x, wt, K, pad_t, pad_l, bn=None, gate_alpha=None, gate_mode=0, add=None, sacc=None):
    """Data gradient AND weight gradient of a stride-1 'same' depthwise conv in one pass over (dy, x) (csrc/dwtile.hip,
    dw_tile_bwd_kernel): da = gate * dwconv_bwd_data(dy) [+ add]; with bn (the deferred BatchNorm in front of the conv, x its
    RAW input): dz = da * act'(bn(x)), sacc += the BatchNorm backward sums of dz, and dw[C, K*K] = gate * sum act(bn(x))(window) * dy;
    without bn: dz = da, the conv's input is x itself.  Returns (dz, dw)."""
    h = _act(dy, x, add)
    _chk(wt)
    N, H, W, Cc = x.shape
    assert dy.shape == x.shape
    rows = _call("ud_dwtile_wgrad_part_rows", N, H, W)
    part, defer = _wgrad_part(x, rows * K * K * Cc)
    dz = empty((N, H, W, Cc), dy, dy.dtype)
    dwt = empty((Cc, K * K), x)
    ws = _ws64(dy, _call("ud_dwtile_ws_doubles", N, H, W, Cc)) if bn is not None else None
    nparts = _call("ud_dwtile_bwd", _p(dy), _p(x), _bnp(bn), _p(wt), _p(gate_alpha), int(gate_mode), _p(add), _p(dz),
                   None if defer else _p(dwt), _p(part), rows, _pd(sacc) if bn is not None else None,
                   _pd(sacc, Cc) if bn is not None else None, ws, N, H, W, Cc, K, pad_t, pad_l, h, _stream())
    if defer:
        _defer_wgrad_fold(part, nparts, K, Cc, gate_alpha, gate_mode, dwt)
    return dz, dwt


_IRFFT_DWBWD = True          # A/B: tools/run_with.py kernels._IRFFT_DWBWD=False
_IRFFT_DWBWD_SIZES = (8, 16)
_IRFFT_DWBWD_HALF = True          # ... also with half-stored tensors (the mixed-precision mode); A/B: =False
# weight gradient of ud_irfft2_dwbwd by fp32 atomics onto the parameter-layout gradient instead of partial rows + the fold launch:
# OFF — 1600 device-scope atomics per workgroup make the 8 x 8 kernel 81-92 us instead of 29 (16 x 16, k 5: 63 vs 52); the step
# with the 18 fold launches is 25.93 ms against 26.28 (profiles/r05/dwbwd_wgrad_atomics_ab.txt; A/B: tools/run_with.py
# kernels._IRFFT_DWBWD_ATOMIC=True)
_IRFFT_DWBWD_ATOMIC = False


def irfft2_dwbwd_ok(S, k, stride, pad, dtype):
    """the SF block's spatial-branch backward inside the adjoint transform (csrc/fft.hip: irfft2_dwbwd_kernel): the 8 x 8 maps"""
    return (_IRFFT_DWBWD and S in _IRFFT_DWBWD_SIZES and k in (3, 5) and stride == 1 and tuple(pad) == ((k - 1) // 2,) * 4
            and (dtype == torch.float32 or (_IRFFT_DWBWD_HALF and dtype == torch.float16)))


def irfft2_dwbwd(Y, scale, w_interior, dd, x, bn, wt, k, gate_alpha, gate_mode, sacc):
    """da_f = irfft2(Y) (the adjoint of rfft2), dz = (gate * dwconv_bwd_data(dd) + da_f) * act'(bn(x)), sacc += BatchNorm backward
    sums of dz (a 3C accumulator: + its energy), dw[C, k*k] = gate * sum act(bn(x))(window) * dd — ONE kernel over the (n, c)
    planes + the partials' fold.  Returns (dz, dw)."""
    h = _act(Y, dd, x)
    _chk(wt)
    N, S, Wh, C2 = Y.shape
    Cc = C2 // 2
    assert dd.shape == (N, S, S, Cc) and x.shape == dd.shape
    dz = torch.empty_like(dd)
    en = _pd(sacc, 2 * Cc) if sacc.numel() >= 3 * Cc else None          # a 3C accumulator: + sum dz^2 (normbwd_apply_planes' bound)
    if _IRFFT_DWBWD_ATOMIC and not CFG.deterministic:
        # the weight gradient by fp32 atomics onto a zeroed [C, k*k] (N adds per address): no partial rows, no fold launch
        dwt = zeros((Cc, k * k), x)
        _call("ud_irfft2_dwbwd", _p(Y), N, S, Cc, float(scale), float(w_interior), _p(dd), _p(x), C.byref(bn.ref()), _p(wt), int(k),
              _p(gate_alpha), int(gate_mode), _p(dz), _pd(sacc), _pd(sacc, Cc), en, None, _p(dwt), h, _stream())
        return dz, dwt
    part, defer = _wgrad_part(x, N * k * k * Cc)
    dwt = empty((Cc, k * k), x)
    _call("ud_irfft2_dwbwd", _p(Y), N, S, Cc, float(scale), float(w_interior), _p(dd), _p(x), C.byref(bn.ref()), _p(wt), int(k),
          _p(gate_alpha), int(gate_mode), _p(dz), _pd(sacc), _pd(sacc, Cc), en, _p(part), None, h, _stream())
    if defer:
        _defer_wgrad_fold(part, N, int(k), Cc, gate_alpha, gate_mode, dwt)
    else:
        _call("ud_dwtile_wgrad_finalize", _p(part), N, int(k), Cc, _p(gate_alpha), int(gate_mode), _p(dwt), _stream())
    return dz, dwt


# ---- k x k convs as 1x1 convs on the planes GEMM: im2col written as planes (csrc/gemm_p3.hip: im2col_planes_kernel) ---------------
_CONV_IM2COL = True                  # A/B: tools/run_with.py kernels._CONV_IM2COL=False
_CONV_IM2COL_MIN_K = 1152            # reduction length KH*KW*Cin from which the im2col form is taken
_CONV_IM2COL_MAX_BYTES = 768 << 20   # ... and the largest im2col matrix (4 bytes per element as planes)
_CONV_IM2COL_MIN_FLOP = 2e9          # 2 M K N below which the conv stays on the gather GEMM (ONE launch).  With the paired backward launches: UDR18 bs 8 (2.4 GFLOP per conv) 4.81 -> 4.70 ms, bs 32 9.1 -> 8.95, UDR50 and the UDEB4 decoder (3.8) unchanged; at 5e9 before the pairs (UDR18 slower as im2col then)


def conv_im2col_ok(g, Co, x):
    """does this conv (ud_conv_geom g, Co output channels) run as im2col planes + ud_gemm_p3?  fp32, F.conv2d geometry, whole
    32-channel panels per tap, a deep reduction (the shallow ones are HBM-bound: a 9x larger operand loses), pixel count a
    multiple of 32 (whole K-tiles for the weight gradient)."""
    M, Kc = g.N * g.Hout * g.Wout, g.KH * g.KW * g.Cin
    return (_CONV_IM2COL and x.dtype == torch.float32 and not g.transposed and g.KH * g.KW > 1 and g.Cin % 32 == 0 and
            Kc >= _CONV_IM2COL_MIN_K and 4 * M * Kc <= _CONV_IM2COL_MAX_BYTES and 2.0 * M * Kc * Co >= _CONV_IM2COL_MIN_FLOP and
            CFG.spectral_p2 != "off" and
            _p2_shape_ok(M, Co, Kc) and _call("ud_gemm_get_path") in (0, 2))


def im2col_planes(x, g, absmax=None):
    """the conv's im2col matrix [N Hout Wout, KH KW Cin] as prec-2 Planes (scale from |x|max: `absmax` slots or a pass here)"""
    _chk(x)
    M, Kc = g.N * g.Hout * g.Wout, g.KH * g.KW * g.Cin
    if absmax is None:
        absmax = empty((256,), x)
        x2 = x.view(-1, g.Cin)
        _call("ud_absmax", _p(x2), x2.shape[0], g.Cin, g.Cin, _p(absmax), _stream())
    pl = Planes(M, Kc, x, 2, False)
    _call("ud_im2col_planes", _p(x), C.byref(g), _p(pl.buf), pl.panel, pl.plane, _p(absmax), _p(pl.inv), _stream())
    return pl


def col2im(dcol, g):
    _chk(dcol)
    dx = empty((g.N, g.Hin, g.Win, g.Cin), dcol)
    _call("ud_col2im", _p(dcol), C.byref(g), _p(dx), _stream())
    return dx

