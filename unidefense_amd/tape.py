"""Reverse-mode tape + differentiable operators of the UniDefense hot path, built on the HIP kernels.

The whole network is ONE node in torch's autograd graph (``model.unidefense._NetFunction``): its forward
runs these operators while recording backward closures on a ``Tape``; its backward replays the tape in
reverse.  Compared with ~2000 ``torch.autograd.Function`` nodes per step this keeps the host path short
and gives the executor full control over buffer lifetime and gradient accumulation.

Every operator takes pixel-major fp32 CUDA tensors (see kernels.py) and cites the reference op it replaces.
"""
import math
from typing import Optional

import torch

from . import kernels as K
from .config import cfg


class Tape:
    """Records backward closures; gradients are keyed by tensor identity."""

    def __init__(self):
        self.nodes = []
        self.grads = {}
        self.param_grads = {}
        self._keep = []          # keep keyed tensors alive so ids stay unique
        self.watch = None        # debug: {id(tensor): name} -> gradients captured into self.captured
        self.captured = {}
        self.kinks = None        # parity tests: {site: ReLU output} (site = id(norm weight) or an explicit name)
        self.gate_cond = {}      # parity tests (watch set): {id(sf_coef): sigmoid'(a) * sum |dy| |freq - spat|}, the
        #                          conditioning of that scalar gradient (a global sum that cancels heavily)
        # data parallel (engine/parallel.py): param_ready(id(p), g) is called the moment p's gradient is final, i.e.
        # after its param_uses[id(p)]-th contribution (counts learned from the first backward: param_seen)
        self.param_ready = None
        self.param_uses = None
        self.param_seen = {}
        self._deferred = set()   # id(param) whose stored gradient still waits for the one-launch fold (kernels.flush_wgrad_folds)
        self.weight_snapshot = None      # kernels.weight_batch_snapshot() of the forward (checked before the replay)
        # side branch (cfg.side_branch): nodes recorded inside `side_branch(tape, ...)` — the reconstruction decoder and its loss
        # tail, independent of the trunk's stage 5 / attention / stage 6 / head — replay on a second stream
        self.side = None         # torch.cuda.Stream of the branch, or None
        self.side_nodes = set()  # id(fn) of the closures recorded on it
        self._in_side = False
        self._side_param_grads = []

    # -- recording -------------------------------------------------------------------------
    def record(self, fn):
        self.nodes.append(fn)
        if self._in_side:
            self.side_nodes.add(id(fn))

    def add_grad(self, t: torch.Tensor, g: torch.Tensor):
        """Accumulate g into the gradient slot of activation t (functional: never mutates g)."""
        k = id(t)
        cur = self.grads.get(k)
        if cur is None:
            self.grads[k] = g
            self._keep.append(t)
        else:
            s = K.axpby(cur, 1.0, g.reshape(cur.shape), 1.0)
            s._ud_owned = True          # a fresh tensor only the tape holds: its consumer may accumulate onto it in place
            self.grads[k] = s

    def pop_grad(self, t: torch.Tensor) -> Optional[torch.Tensor]:
        g = self.grads.pop(id(t), None)
        if self.watch is not None and id(t) in self.watch:
            self.captured[self.watch[id(t)]] = g
        return g

    def add_param_grad(self, p, g: torch.Tensor):
        cur = self.param_grads.get(p)
        deferred = getattr(g, "_ud_deferred", False)
        if (deferred or id(p) in self._deferred) and (cur is not None or self.param_ready is not None):
            K.flush_wgrad_folds()          # read right away (summed with an earlier use / handed to the gradient reducer): fold now
            self._deferred.clear()         # (the flush folds every pending gradient, the stored ones included)
            deferred = False
        g = g.reshape(p.shape)
        g = g if cur is None else K.axpby(cur, 1.0, g, 1.0)
        self.param_grads[p] = g
        if self._in_side:
            self._side_param_grads.append(g)
        if deferred:
            self._deferred.add(id(p))      # stored unfolded: a later contribution to p must flush before it reads `cur`
        n = self.param_seen[id(p)] = self.param_seen.get(id(p), 0) + 1
        if self.param_ready is not None and n == self.param_uses.get(id(p)):
            self.param_ready(id(p), g)

    # -- weight gradients -------------------------------------------------------------------
    def wgrad(self, p, fn, *inputs):
        """Run `fn()` (the weight-gradient kernels of parameter p) and record the result.  One queue: a second stream for
        these launches was measured slower in rounds 1, 2 and 4 (46.3 vs 43.6, 36.15 vs 34.70, 29.1 vs 27.2 ms per step: the
        GEMMs own the CU's registers and LDS, nothing co-schedules) and is gone — with it the ordering hazard of operand
        planes made lazily on whichever stream touched them first."""
        self.add_param_grad(p, fn())

    # -- replay ----------------------------------------------------------------------------
    def backward(self):
        if self.weight_snapshot is not None:
            K.weight_batch_check(self.weight_snapshot)
        K.begin_wgrad_folds()          # the depthwise weight gradients' folds: one launch at the end instead of one per conv
        try:
            if self.side is None or not self.side_nodes:
                for fn in reversed(self.nodes):
                    fn()
            else:
                self._backward_two_streams()
        finally:
            K.flush_wgrad_folds(end=True)
        self.nodes = []
        self.grads = {}
        self._keep = []
        self.side_nodes = set()

    def _backward_two_streams(self):
        """The replay with the side branch's closures on its stream.  The recorded order is a valid serial order, so each stream
        runs its closures in that order; the branch starts (waits for the main stream) at its first closure — everything it
        reads from outside are the output gradients, set before the replay — and the main stream waits for it before the first
        closure recorded BEFORE the branch began (the consumer of the branch input's gradient).  Closures of the main stream in
        between (stage 5 ... head) exchange nothing with the branch: the attention reads the reconstruction detached."""
        main, side = torch.cuda.current_stream(), self.side
        side_ids = self.side_nodes
        first = min(i for i, fn in enumerate(self.nodes) if id(fn) in side_ids)
        started = joined = False
        for i in range(len(self.nodes) - 1, -1, -1):
            fn = self.nodes[i]
            if id(fn) in side_ids:
                if not started:
                    for g in self.grads.values():          # output gradients made on the main stream, read on the branch
                        g.record_stream(side)
                    side.wait_stream(main)
                    started = True
                with torch.cuda.stream(side), K.branch(1):
                    self._in_side = True
                    try:
                        fn()
                    finally:
                        self._in_side = False
            else:
                if started and not joined and i < first:
                    self._join(main, side)
                    joined = True
                fn()
        if started and not joined:
            self._join(main, side)

    def _join(self, main, side):
        for g in self.grads.values():                      # the branch input's gradient: made on the branch, read on main
            g.record_stream(main)
        for g in self._side_param_grads:
            g.record_stream(main)
        self._side_param_grads = []
        main.wait_stream(side)


def _needs(tape):
    return tape is not None


_SIDE_STREAMS = {}


class side_branch:
    """`with side_branch(tape, like, inputs): ...` — the enclosed operators run on a second stream, concurrently with whatever
    the caller launches on the main stream until `.join(outputs)`; their backward closures replay on that stream too
    (Tape.backward).  Used for the reconstruction decoder (model/unidefense.py:214-216 of the reference), which shares only its
    input with the trunk's stage 5: the two are latency-bound chains of small kernels on different tensors.  `inputs`: tensors
    made on the main stream that the branch reads (their memory must not be recycled under it).  Off (plain serial execution)
    without a tape, with cfg.side_branch = False, or on a tape that already carries a branch."""

    def __init__(self, tape, like, inputs=()):
        self.tape = tape
        self.on = bool(tape is not None and cfg.side_branch and like.is_cuda and tape.side is None)
        self.like, self.inputs = like, inputs
        self.ctx = None

    def __enter__(self):
        if not self.on:
            return self
        dev = self.like.device.index
        side = _SIDE_STREAMS.get(dev)
        if side is None:
            side = _SIDE_STREAMS[dev] = torch.cuda.Stream(device=self.like.device)
        self.main = torch.cuda.current_stream()
        for t in self.inputs:
            t.record_stream(side)
        side.wait_stream(self.main)
        self.tape.side = side
        self.tape._in_side = True
        self.ctx = (torch.cuda.stream(side), K.branch(1))
        self.ctx[0].__enter__()
        self.ctx[1].__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx[1].__exit__(*exc)
            self.ctx[0].__exit__(*exc)
            self.tape._in_side = False
        return False

    def join(self, outputs=()):
        """the main stream waits for the branch; `outputs`: tensors the branch made that the main stream goes on to read"""
        if self.ctx is None:
            return
        for t in outputs:
            if t is not None:
                t.record_stream(self.main)
        self.main.wait_stream(self.tape.side)
        self.ctx = None


# ---------------------------------------------------------------------------------------------
# GEMM-shaped ops
# ---------------------------------------------------------------------------------------------
_CONST_AFFINE = {}


def _const_affine(Cc, like):
    """gamma = 1, beta = 0 for a norm built with affine=False (model/unidefense.py:38,61,116): constants, no gradients"""
    key = (Cc, like.device)
    v = _CONST_AFFINE.get(key)
    if v is None:
        v = _CONST_AFFINE[key] = (torch.ones(Cc, device=like.device), torch.zeros(Cc, device=like.device))
    return v


def bias_add(tape, y, b):
    """+ the bias of a conv built with bias=True (model/unidefense.py:36,60-100; modules.py:82,87,111,116) on channel-last y.
    No shipped config of the reference uses the variant, so these are plain device ops: a broadcast add, and a column sum for
    the bias gradient."""
    if b is None:
        return y
    out = y + b
    if _needs(tape):
        def bwd():
            d = tape.pop_grad(out)
            if d is None:
                return
            tape.add_grad(y, d)
            tape.add_param_grad(b, d.reshape(-1, d.shape[-1]).sum(0))
        tape.record(bwd)
    return out


def conv1x1(tape, x, w, need_dx=True):
    """F.conv2d with a [Cout,Cin,1,1] weight on pixel-major x[...,Cin] (model/efficientnet/model.py:108,125;
    exp.py:57; model/modules.py:82).  Backward = convolution_backward: dX = dY W, dW = dY^T X."""
    Co, Ci = w.shape[0], w.shape[1]
    w2 = w.view(Co, Ci)
    x2 = x.view(-1, Ci)
    if not _needs(tape):
        return K.gemm_nt(x2, w2).view(*x.shape[:-1], Co)
    # the three products share their operands: large shapes split them once into fp16 x 2 planes (kernels.spectral_*)
    y2, ctx = K.spectral_fwd(x2, w2)
    y = y2.view(*x.shape[:-1], Co)

    def bwd():
        dy = tape.pop_grad(y)
        if dy is None:
            return
        dy2 = dy.reshape(-1, Co)
        if need_dx:
            dx, dw = K.spectral_bwd(ctx, dy2)          # (one launch on the planes GEMM where both plans allow)
            tape.add_param_grad(w, dw)
            tape.add_grad(x, dx.view(x.shape))
        else:
            tape.wgrad(w, lambda: K.spectral_wgrad(ctx, dy2), dy2)
    tape.record(bwd)
    return y


def _conv_im2col(tape, x, w, wmat, g, need_dx):
    """A k x k conv as a 1x1 conv over its im2col matrix on the planes GEMM (kernels.im2col_planes): forward, weight gradient and
    data gradient (a GEMM + col2im) share the three operands' planes like tape.conv1x1's."""
    Co, Ci, KH, KW = w.shape
    xcol = K.im2col_planes(x, g)
    y2, ctx = K.spectral_fwd(xcol, wmat, force=True)
    y = y2.view(g.N, g.Hout, g.Wout, Co)
    if _needs(tape):
        def bwd():
            dy = tape.pop_grad(y)
            if dy is None:
                return
            dy2 = dy.reshape(-1, Co)
            if not dy2.is_contiguous():
                dy2 = dy2.contiguous()
            if need_dx:
                dcol, dw = K.spectral_bwd(ctx, dy2)                    # dw [Co, KH*KW*Ci]
            else:
                dcol, dw = None, K.spectral_wgrad(ctx, dy2)
            tape.add_param_grad(w, dw.view(Co, KH, KW, Ci).permute(0, 3, 1, 2).contiguous())
            if need_dx:
                tape.add_grad(x, K.col2im(dcol, g))
        tape.record(bwd)
    return y


def conv_dense(tape, x, w, stride, pad_t, pad_l, Hout, Wout, need_dx=True):
    """Dense k x k F.conv2d (weight [Cout,Cin,kh,kw]) as an implicit GEMM (model/unidefense.py:60,67,...;
    model/modules.py:111; stem conv model/efficientnet/model.py:185 with its static asymmetric pad)."""
    N, Hin, Win, Ci = x.shape
    Co, _, KH, KW = w.shape
    g = K.conv_geom(N, Hin, Win, Ci, Hout, Wout, KH, KW, stride, pad_t, pad_l, 0)
    wmat = K.weight_layout(w, 0)
    if K.conv_im2col_ok(g, Co, x):
        return _conv_im2col(tape, x, w, wmat, g, need_dx)
    y = K.conv_gather_nt(x, wmat, g)
    if _needs(tape):
        # dX = conv(dY, W flipped & transposed), pad = k-1-pad: its weight matrix comes out of the forward's one-launch batch
        wd = K.weight_layout(w, 1) if need_dx else None

        def bwd():
            dy = tape.pop_grad(y)
            if dy is None:
                return
            if need_dx:
                assert stride == 1 and Hout == Hin and Wout == Win, "data gradient implemented for stride 1 'same'"
                gd = K.conv_geom(N, Hout, Wout, Co, Hin, Win, KH, KW, 1, KH - 1 - pad_t, KW - 1 - pad_l, 0)
                tape.add_grad(x, K.conv_gather_nt(dy, wd, gd))
            dw = K.conv_gather_wgrad(dy.view(-1, Co), x, g)            # [Co, KH*KW*Ci]
            tape.add_param_grad(w, dw.view(Co, KH, KW, Ci).permute(0, 3, 1, 2).contiguous())
        tape.record(bwd)
    return y


def conv_transpose_s2(tape, x, w):
    """nn.ConvTranspose2d(k=3, stride=2, padding=1, output_padding=1), weight [Cin,Cout,3,3]
    (model/unidefense.py:63-64,77-78,91-92)."""
    N, H, W, Ci = x.shape
    _, Co, KH, KW = w.shape
    Ho, Wo = 2 * H, 2 * W
    g = K.conv_geom(N, H, W, Ci, Ho, Wo, KH, KW, 2, 1, 1, 1)
    wmat = K.weight_layout(w, 2)
    y = K.conv_gather_nt(x, wmat, g)
    if _needs(tape):
        wd = K.weight_layout(w, 0)

        def bwd():
            dy = tape.pop_grad(y)
            if dy is None:
                return
            # dX[n,ih,iw,ci] = sum dY[n,2ih-1+kh,2iw-1+kw,co] W[ci,co,kh,kw]: a stride-2 conv over dY
            gd = K.conv_geom(N, Ho, Wo, Co, H, W, KH, KW, 2, 1, 1, 0)
            tape.add_grad(x, K.conv_gather_nt(dy, wd, gd))
            dw = K.conv_gather_wgrad(x.view(-1, Ci), dy, gd)           # [Ci, KH*KW*Co]
            tape.add_param_grad(w, dw.view(Ci, KH, KW, Co).permute(0, 3, 1, 2).contiguous())
        tape.record(bwd)
    return y


def linear(tape, x, w, b):
    """nn.Linear (model/modules.py:27,31)."""
    y = K.fc_fwd(x, w, b, 0)
    if _needs(tape):
        def bwd():
            dy = tape.pop_grad(y)
            if dy is None:
                return
            dx, dW, db = K.fc_bwd(dy, w, x, 0)
            tape.add_grad(x, dx)
            tape.add_param_grad(w, dW)
            tape.add_param_grad(b, db)
        tape.record(bwd)
    return y


# ---------------------------------------------------------------------------------------------
# depthwise conv, FFT, SFConv
# ---------------------------------------------------------------------------------------------
_DW_FUSED_ADD = True
# Stride-1 depthwise convs of the fused MBConv node on the LDS-tiled kernels of csrc/dwtile.hip (the deferred BatchNorm is
# applied while the halo tile is staged: swish(bn0(e)) is never materialised) WHERE THEY WIN — measured per shape against
# the strip kernels with tools/bench_dwtile.py on an MI355X (profiles/r03/dwtile_f32_bs32.txt, dwtile_f16_bs64.txt):
#   half storage  : everywhere, by 1.3 - 3x (forward 109 -> 53, data gradient 260 -> 86, weight gradient 284 -> 106 us ...:
#                   the strip kernels' per-tap 8-byte window loads and conversions);
#   fp32 forward  : every block (one launch replaces bn_apply + conv + colstats of the plain blocks: 118 -> 74, 63 -> 43,
#                   121 -> 52, 37 -> 23 us; SF blocks 43 -> 33, 31 -> 23 us) but the 5x5 / 8x8 maps (16 vs 14 us);
#   fp32 data grad: 5x5 at 16x16 and 32x32 (70 -> 49, 45 -> 36 us) and 3x3 at 32x32 ... 64x64 (101 -> 94 us); the strip kernel
#                   keeps the 8x8 and 128x128 maps;
#   fp32 weight grad: the plain blocks (63 -> 54, 66 -> 55 us at 128^2 / 64^2; they have no materialised activation any
#                   more); the SF blocks keep the strip kernel (36 vs 45, 23 vs 29 us) on the activation rfft2_ex writes.
_DW_TILED = True


# per-shape overrides of the two depthwise policies below, {(sf, k, stride, H, half): value}: measured IN the replayed step by
# tools/tune_in_step.py --knobs (the rules come from per-kernel timings); empty = the rules
_DW_TILE_OVERRIDE = {}
_DW_BWD_FUSED_OVERRIDE = {}


def _dw_tile_policy(sf, k, stride, H, half):
    """(forward, weight gradient, data gradient) on the tiled kernels?"""
    ov = _DW_TILE_OVERRIDE.get((bool(sf), k, stride, H, bool(half)))
    if ov is not None and _DW_TILED:
        return ov
    return _dw_tile_rule(sf, k, stride, H, half)


def _dw_tile_rule(sf, k, stride, H, half):
    if not _DW_TILED:
        return False, False, False
    if stride == 2:
        # the four down-sampling blocks (profiles/r03/dwtile_stride2.txt).  Data gradient through the zero-stuffed tile, with
        # the BatchNorm sums in its epilogue instead of a pass of their own: 189 -> 88, 59 -> 40, 57 -> 32 us (fp32), but not
        # on the 128 x 128 map in half storage (630 -> 699); forward 271 -> 130 us on the plain 128 x 128 block (incl. the
        # BatchNorm apply + statistics passes it absorbs), otherwise only on the large maps; weight gradient: half storage
        # (515 -> 177, 208 -> 100 us) and the plain block, the strip kernel elsewhere (49 vs 63 us)
        if half:
            return True, True, H < 128
        return (not sf) or H >= 64, not sf, True
    if half:
        return True, True, True
    fwd = not (k == 5 and H <= 8)
    bwd = (k == 5 and H >= 16) or (k == 3 and 32 <= H <= 64)
    return (fwd or not sf), not sf, bwd


# Round 5: the stride-1 blocks' depthwise data gradient and weight gradient as ONE kernel (kernels.dwtile_bwd: both halo tiles
# staged once; the separate kernels read dy and the conv's input twice each).  Where it is used: _dw_bwd_fused_policy.
_DW_BWD_FUSED = True


def _dw_bwd_fused_policy(sf, k, stride, H, half):
    """data + weight gradient of this depthwise conv in one launch (csrc/dwtile.hip: dw_tile_bwd_kernel)?  Measured per shape
    against the pair of kernels _dw_tile_policy picks (tools/bench_dwbwd.py on an MI355X, profiles/r05/dwbwd_*.txt; us, pair ->
    fused): fp32 bs 32: 128^2 187 -> 151 and 74 -> 61, 32^2 k5 100 -> 81, 16^2 k5 70 -> 63, 16^2 k3 47 -> 45, 8^2 k3 47 -> 45;
    NOT the 64^2 k3 blocks (132 vs 134: two halo tiles of a 16 x 8 tile are 1.4x its pixels, twice) and NOT the 8^2 k5 blocks
    (48 vs 60: one workgroup per CU next to the strip kernels' eight).  Half storage bs 64: 328 -> 260, 90 -> 80, 224 -> 211,
    163 -> 156, 73 -> 71; a tie at 16^2 k5 and 8^2 k3, 72 vs 96 at 8^2 k5."""
    if not _DW_BWD_FUSED or stride != 1:
        return False
    ov = _DW_BWD_FUSED_OVERRIDE.get((bool(sf), k, stride, H, bool(half)))
    if ov is not None:
        return ov
    if H <= 8:
        return k == 3 and not half
    if half:
        return True
    return not (k == 3 and H == 64)


DW_WT = {}            # {id(w): (w, w._version, tap-major wt)} for the forward in flight (kernels.dw_weights_tapmajor)


def dwconv(tape, x, w, stride, pad):
    """Depthwise Conv2dStaticSamePadding (model/efficientnet/utils.py:277-280; exp.py:49-51).
    pad = (left, right, top, bottom) as in nn.ZeroPad2d."""
    N, H, W, Cc = x.shape
    k = w.shape[-1]
    pl, pr, pt, pb = pad
    Ho = (H + pt + pb - k) // stride + 1
    Wo = (W + pl + pr - k) // stride + 1
    ent = DW_WT.get(id(w))                      # tap-major copy made for all layers at once (model._run), if any
    if ent is not None and ent[0] is w and ent[1] == w._version:
        wt = ent[2]
    else:
        wt = w.view(Cc, k * k).t().contiguous()
    y = K.dwconv_fwd(x, wt, k, stride, pt, pl, Ho, Wo)
    if _needs(tape):
        def bwd():
            dy = tape.pop_grad(y)
            if dy is None:
                return
            # x usually has a second consumer whose gradient is already there (SFConv's frequency branch, exp.py:55;
            # the SE pool): the data-gradient kernel adds it in its store instead of a separate axpby pass
            cur = tape.grads.pop(id(x), None) if _DW_FUSED_ADD else None
            if cur is not None and not (cur.is_contiguous() and cur.shape == x.shape and cur.dtype == torch.float32):
                tape.grads[id(x)], cur = cur, None
            tape.add_grad(x, K.dwconv_bwd_data(dy, wt, k, stride, pt, pl, H, W, add=cur))
            tape.add_param_grad(w, K.dwconv_bwd_weight(x, dy, k, stride, pt, pl))     # already [C, k*k]
        tape.record(bwd)
    return y


def _fft_scales(S, norm):
    if norm == "ortho":
        return 1.0 / S, 1.0 / S
    if norm is None or norm == "backward":
        return 1.0, 1.0 / (S * S)
    raise ValueError(f"unsupported fft norm {norm!r}")


def rfft2_cat(tape, x, norm):
    """torch.fft.rfft2 + cat([re, im], channel)  (exp.py:55-56; unidefense.py:130-136)."""
    S = x.shape[1]
    sf, _ = _fft_scales(S, norm)
    y = K.rfft2(x, sf, 1.0)
    if _needs(tape):
        def bwd():
            dy = tape.pop_grad(y)
            if dy is None:
                return
            tape.add_grad(x, K.irfft2(dy, sf, 0.5))
        tape.record(bwd)
    return y


def irfft2_split(tape, y, norm):
    """tensor_split + torch.complex + torch.fft.irfft2(s=(S,S))  (exp.py:59-60; unidefense.py:142-145)."""
    S = y.shape[1]
    _, si = _fft_scales(S, norm)
    x = K.irfft2(y, si, 1.0)
    if _needs(tape):
        def bwd():
            dx = tape.pop_grad(x)
            if dx is None:
                return
            tape.add_grad(y, K.rfft2(dx, si, 2.0))
        tape.record(bwd)
    return x


def adaptive_avgpool(tape, x, Ho, Wo):
    """F.adaptive_avg_pool2d to a size that does not divide the input (exp.py:61-62 on the 95 x 95 map of the 380 x 380 trunk's
    stride-2 SF block: 95 -> 48, windows of 2 and 3 that overlap): csrc/pool.hip's adaptive_avgpool kernels (rounds 3-4: ATen's);
    the 2 x 2 case of the 256 x 256 trunk is fused into ud_sfmix."""
    H, W = x.shape[1], x.shape[2]
    y = K.adaptive_avgpool_fwd(x, Ho, Wo)
    if _needs(tape):
        def bwd():
            dy = tape.pop_grad(y)
            if dy is None:
                return
            tape.add_grad(x, K.adaptive_avgpool_bwd(dy.contiguous(), H, W))
        tape.record(bwd)
    return y


def sfmix(tape, spat, freq, alpha):
    """(1 - sigmoid(a)) * spat + sigmoid(a) * [avg-pooled] freq  (exp.py:61-65)."""
    if freq.shape[1] != spat.shape[1] and (freq.shape[1] != 2 * spat.shape[1] or freq.shape[2] != 2 * spat.shape[2]):
        freq = adaptive_avgpool(tape, freq, spat.shape[1], spat.shape[2])
    pool = freq.shape[1] != spat.shape[1]
    if pool:
        assert freq.shape[1] == 2 * spat.shape[1] and freq.shape[2] == 2 * spat.shape[2]
    y = K.sfmix_fwd(spat, freq, alpha, pool)
    if _needs(tape):
        def bwd():
            dy = tape.pop_grad(y)
            if dy is None:
                return
            ds, df, da = K.sfmix_bwd(spat, freq, alpha, dy, pool)
            if tape.watch is not None:
                with torch.no_grad():
                    f = freq.float()
                    if pool:
                        n_, h_, w_, c_ = spat.shape
                        f = f.view(n_, h_, 2, w_, 2, c_).mean((2, 4))
                    sg = torch.sigmoid(alpha.double())
                    tape.gate_cond[id(alpha)] = float(sg * (1 - sg) * (dy.double().abs() * (f.double() - spat.double()).abs()).sum())
            tape.add_grad(spat, ds)
            tape.add_grad(freq, df)
            tape.add_param_grad(alpha, da)
        tape.record(bwd)
    return y


def sfconv_dw(tape, x, w, w_freq, alpha, stride, pad, norm):
    """SFConv2dStaticSamePadding.forward (model/efficientnet/exp.py:46-65)."""
    spat = dwconv(tape, x, w, stride, pad)
    xf = rfft2_cat(tape, x, norm)
    yf = conv1x1(tape, xf, w_freq)
    fr = irfft2_split(tape, yf, norm)
    return sfmix(tape, spat, fr, alpha)


# ---------------------------------------------------------------------------------------------
# normalisation + activation
# ---------------------------------------------------------------------------------------------
# cfg.force_collectives: issue the data-parallel collectives (SyncBN statistics, gradient all-reduce) even in a
# world of one process, so that a single-GPU box exercises the RCCL calls and their hipGraph capture


def sync_batch_stats(mean_l, var_l, eps, group):
    """Combine per-rank (mean, biased var) over equally sized shards into the global statistics:
    mean = avg_r mean_r ;  var = avg_r (var_r + (mean_r - mean)^2).  One all_gather of 2C floats
    (torch.nn.SyncBatchNorm's forward exchange; RCCL on GPU, gloo in the CPU tests)."""
    import torch.distributed as dist
    ws = dist.get_world_size(group)
    mine = torch.cat([mean_l.reshape(-1), var_l.reshape(-1)])
    flat = torch.empty(ws * mine.numel(), dtype=mine.dtype, device=mine.device)
    dist.all_gather_into_tensor(flat, mine, group=group)
    allst = flat.view(ws, mine.numel())
    Cc = mean_l.numel()
    means, vars_ = allst[:, :Cc], allst[:, Cc:]
    mean = means.mean(0)
    var = (vars_ + (means - mean) ** 2).mean(0)
    return mean.view(1, Cc).contiguous(), var.view(1, Cc), torch.rsqrt(var + eps).view(1, Cc).contiguous()


def batchnorm_act(tape, x, weight, bias, running_mean, running_var, eps, momentum, training, act, sync_group=None,
                  exchange=None):
    """nn.BatchNorm2d/1d (+ MemoryEfficientSwish when act=1) on pixel-major x[..., C]
    (model/efficientnet/model.py:109-114,126; utils.py:66-82).  With sync_group: SyncBatchNorm semantics
    (statistics over the batches of all ranks; engine/forgery_engine.py:142, ocim_engine.py:130-133) — the fp64 sums
    (sum x, sum x^2; sum dz, sum dz xhat in the backward) travel through `exchange` (engine.parallel.BnExchange: one
    small kernel over peer-mapped mailboxes) when given, else through one dist.all_reduce each way."""
    Cc = x.shape[-1]
    x2 = x.view(-1, Cc)
    R = x2.shape[0]
    affine = weight is not None
    if not affine:
        weight, bias = _const_affine(Cc, x)
    synced = one = False
    if training and sync_group is not None:
        import torch.distributed as dist
        synced = dist.get_world_size(sync_group) > 1 or cfg.force_collectives
    if synced and Cc % 4 == 0:
        return _syncbn_act(tape, x, x2, R, Cc, weight, bias, running_mean, running_var, eps, momentum, act, sync_group,
                           exchange, affine)
    if synced:
        # channel counts the column kernels do not take (never in the reference's models): (mean, var) all_gather form
        world = dist.get_world_size(sync_group)
        mv = K.norm_stats_local(x2, 1, R, eps)                               # [2, 1, C]
        gathered = torch.empty((world, 2, Cc), dtype=mv.dtype, device=mv.device)
        dist.all_gather_into_tensor(gathered.view(-1), mv.view(-1), group=sync_group)
        mean, invstd = K.syncbn_combine(gathered, world, Cc, R, eps, momentum, running_mean, running_var)
    elif training:
        world = 1
        one = K.norm_fused_ok(x2, 1, R)          # statistics + apply in one launch (csrc/norm.hip, norm_fwd_fused)
        if one:
            y, mean, invstd = K.norm_fwd_fused(x2, 1, R, weight, bias, act, eps, momentum, running_mean, running_var)
        else:
            mean, invstd = K.norm_stats(x2, 1, R, eps, momentum, running_mean, running_var)
    else:
        world = 1
        mean = running_mean.view(1, Cc)
        invstd = torch.rsqrt(running_var + eps).view(1, Cc)
    if not one:
        y = K.norm_apply(x2, 1, R, mean, invstd, weight, bias, act)
    y = y.view(x.shape)
    if act == 2 and tape is not None and tape.kinks is not None:
        tape.kinks[id(weight)] = y
    if _needs(tape):
        def bwd():
            dy = tape.pop_grad(y)
            if dy is None:
                return
            if not training:
                # eval mode: the statistics are constants (running_mean / running_var), so dx = gamma * invstd * dz with
                # dz = dy * act'(z) — the training formula with its two mean terms dropped (sums passed as zeros) — and
                # dgamma = sum dz * xhat, dbeta = sum dz as in training
                s, dg, db = K.norm_bwd_sums(x2, dy.view(-1, Cc), 1, R, mean, invstd, weight, bias, act)
                dx = K.norm_bwd_apply(x2, dy.view(-1, Cc), 1, R, mean, invstd, weight, bias, torch.zeros_like(s), 0.0, act)
            elif synced:
                import torch.distributed as dist
                s, dg, db = K.norm_bwd_sums(x2, dy.view(-1, Cc), 1, R, mean, invstd, weight, bias, act)
                dist.all_reduce(s, group=sync_group)           # sum_dz, sum_dz_xhat over all ranks
                dx = K.norm_bwd_apply(x2, dy.view(-1, Cc), 1, R, mean, invstd, weight, bias, s,
                                      1.0 / float(R * world), act)
            else:
                dx, dg, db = (K.norm_bwd_fused if one else K.norm_bwd)(x2, dy.view(-1, Cc), 1, R, mean, invstd, weight, bias, act)
            tape.add_grad(x, dx.view(x.shape))
            if affine:
                tape.add_param_grad(weight, dg)
                if bias.requires_grad:
                    tape.add_param_grad(bias, db)
        tape.record(bwd)
    return y


def _syncbn_act(tape, x, x2, R, Cc, weight, bias, running_mean, running_var, eps, momentum, act, sync_group, exchange,
                affine=True):
    """SyncBatchNorm of the operator path (attention, head, the ResNet models) on the deferred-BatchNorm kernels of the
    fused path: fp64 column sums -> summed over the ranks in place (DataParallelCtx.reduce: BnExchange's mailbox kernel or
    one all_reduce) -> one apply pass that also moves the running statistics; the backward sums the same way.  Every rank
    must hold the same number of rows (the count is R * world; engine/data.py pads the shards like DistributedSampler)."""
    dp = DataParallelCtx(sync_group, exchange)
    acc = K.zeros64(2 * Cc, x)
    K.colstats(x2, acc)
    dp.reduce(acc)
    bn = K.DeferredBN(acc, Cc, R * dp.world, weight, bias, eps, act, momentum, running_mean, running_var)
    y = K.bn_apply(x2, bn, 1, R, update=True).view(x.shape)
    if act == 2 and tape is not None and tape.kinks is not None:
        tape.kinks[id(weight)] = y
    if _needs(tape):
        def bwd():
            dy = tape.pop_grad(y)
            if dy is None:
                return
            dy2 = dy.reshape(-1, Cc)
            if not dy2.is_contiguous():
                dy2 = dy2.contiguous()
            sb = K.zeros64(2 * Cc, x)
            K.normbwd_sums(x2, dy2, None, 1.0, bn, False, 1, R, sb)
            loc = dp.reduce(sb, keep_local=True)
            dx, dg, db = K.normbwd_apply(x2, dy2, None, 1.0, bn, False, 1, R, sb, loc, want_dbeta=bias.requires_grad)
            tape.add_grad(x, dx.view(x.shape))
            if affine:
                tape.add_param_grad(weight, dg)
                if bias.requires_grad:
                    tape.add_param_grad(bias, db)
        tape.record(bwd)
    return y


def instancenorm_act(tape, x, weight, bias, eps, act):
    """nn.InstanceNorm2d(affine=affine) (+swish) on x[N,H,W,C]  (model/unidefense.py:61-70); weight = bias = None: affine=False."""
    N, H, W, Cc = x.shape
    x2 = x.view(-1, Cc)
    affine = weight is not None
    if not affine:
        weight, bias = _const_affine(Cc, x)
    one = K.norm_fused_ok(x2, N, H * W)          # statistics + apply in one launch (csrc/norm.hip, norm_fwd_fused)
    if one:
        y, mean, invstd = K.norm_fwd_fused(x2, N, H * W, weight, bias, act, eps)
        y = y.view(x.shape)
    else:
        mean, invstd = K.norm_stats(x2, N, H * W, eps)
        y = K.norm_apply(x2, N, H * W, mean, invstd, weight, bias, act).view(x.shape)
    if act == 2 and tape is not None and tape.kinks is not None:
        tape.kinks[id(weight)] = y
    if _needs(tape):
        def bwd():
            dy = tape.pop_grad(y)
            if dy is None:
                return
            dx, dg, db = (K.norm_bwd_fused if one else K.norm_bwd)(x2, dy.view(-1, Cc), N, H * W, mean, invstd, weight, bias, act)
            tape.add_grad(x, dx.view(x.shape))
            if affine:
                tape.add_param_grad(weight, dg)
                tape.add_param_grad(bias, db)
        tape.record(bwd)
    return y


# ---------------------------------------------------------------------------------------------
# pooling, SE, residual, dropout
# ---------------------------------------------------------------------------------------------
def mean_hw(tape, x):
    """x.mean([-2,-1]) / adaptive_avg_pool2d(x, 1) -> [N, C]  (unidefense.py:226,232-236)."""
    N, H, W, Cc = x.shape
    y = K.group_colsum(x.view(-1, Cc), N, H * W, 1.0 / (H * W))
    if _needs(tape):
        def bwd():
            dy = tape.pop_grad(y)
            if dy is None:
                return
            tape.add_grad(x, K.bcast_rows(dy, H * W, 1.0 / (H * W)).view(x.shape))
        tape.record(bwd)
    return y


def squeeze_excite(tape, x, w_r, b_r, w_e, b_e):
    """model/efficientnet/model.py:117-122."""
    N, H, W, Cc = x.shape
    Cs = w_r.shape[0]
    wr2, we2 = w_r.view(Cs, Cc), w_e.view(Cc, Cs)
    pool = K.group_colsum(x.view(-1, Cc), N, H * W, 1.0 / (H * W))
    s1 = K.fc_fwd(pool, wr2, b_r, 0)
    s2 = K.fc_fwd(s1, we2, b_e, 1)          # swish on the input side
    y = K.se_scale_fwd(x, s2)
    if _needs(tape):
        def bwd():
            dy = tape.pop_grad(y)
            if dy is None:
                return
            ds2 = K.group_coldot(dy.view(-1, Cc), x.view(-1, Cc), N, H * W)
            K.sigmoid_grad_mul_(s2, ds2)
            ds1, dWe, dbe = K.fc_bwd(ds2, we2, s1, 1)
            dpool, dWr, dbr = K.fc_bwd(ds1, wr2, pool, 0)
            tape.add_grad(x, K.se_scale_bwd(dy, s2, dpool))
            tape.add_param_grad(w_e, dWe)
            tape.add_param_grad(b_e, dbe)
            tape.add_param_grad(w_r, dWr)
            tape.add_param_grad(b_r, dbr)
        tape.record(bwd)
    return y


def residual(tape, x, skip, keep=None, keep_prob=1.0):
    """drop_connect + skip: x / keep_prob * keep[n] + skip  (model.py:130-134; utils.py:131-156)."""
    inv = 1.0 / keep_prob
    y = K.residual(x, skip, keep, inv)
    if _needs(tape):
        def bwd():
            dy = tape.pop_grad(y)
            if dy is None:
                return
            tape.add_grad(skip, dy)
            tape.add_grad(x, dy if keep is None else K.residual(dy, None, keep, inv))
        tape.record(bwd)
    return y


def add(tape, a, b):
    y = K.axpby(a, 1.0, b, 1.0)
    if _needs(tape):
        def bwd():
            dy = tape.pop_grad(y)
            if dy is None:
                return
            tape.add_grad(a, dy)
            tape.add_grad(b, dy)
        tape.record(bwd)
    return y


def dropout_mask(tape, x, keep, p):
    """F.dropout with an explicit keep-mask: x * keep / (1 - p)."""
    sc = 1.0 / (1.0 - p)
    y = K.mask_scale(x, keep, sc)
    if _needs(tape):
        def bwd():
            dy = tape.pop_grad(y)
            if dy is None:
                return
            tape.add_grad(x, K.mask_scale(dy, keep, sc))
        tape.record(bwd)
    return y


def gate_mix(tape, p, q, alpha):
    """(1 - sigmoid(a)) p + sigmoid(a) q   (fuse_coef, model/unidefense.py:153-154)."""
    y = K.gate_mix_fwd(p, q, alpha)
    if _needs(tape):
        def bwd():
            dy = tape.pop_grad(y)
            if dy is None:
                return
            dp, dq, da = K.gate_mix_bwd(p, q, alpha, dy)
            if tape.watch is not None:
                with torch.no_grad():
                    sg = torch.sigmoid(alpha.double())
                    tape.gate_cond[id(alpha)] = float(sg * (1 - sg) * (dy.double().abs() * (q.double() - p.double()).abs()).sum())
            tape.add_grad(p, dp)
            tape.add_grad(q, dq)
            tape.add_param_grad(alpha, da)
        tape.record(bwd)
    return y


# ---------------------------------------------------------------------------------------------
# image-domain tail
# ---------------------------------------------------------------------------------------------
def tanh_to_planes(tape, x):
    """nn.Tanh on [N,H,W,3] and move to planes [N,3,H,W]  (model/unidefense.py:101)."""
    y = K.pix_to_planes(x, tanh=True)
    if _needs(tape):
        def bwd():
            dy = tape.pop_grad(y)
            if dy is None:
                return
            tape.add_grad(x, K.planes_to_pix(dy, tanh_out=y))
        tape.record(bwd)
    return y


def bilinear(tape, x, Ho, Wo):
    """F.interpolate(mode='bilinear', align_corners=True) on planes (model/unidefense.py:16,244)."""
    Hi, Wi = x.shape[-2:]
    y = K.bilinear_fwd(x, Ho, Wo)
    if _needs(tape):
        def bwd():
            dy = tape.pop_grad(y)
            if dy is None:
                return
            tape.add_grad(x, K.bilinear_bwd(dy, Hi, Wi))
        tape.record(bwd)
    return y


def rec_losses(tape, rec, x, norm):
    """spatial = mean|rec - x| ; freq = mean(|Re D| + |Im D|), D = rfft2(rec) - rfft2(x) = rfft2(rec - x)
    (model/unidefense.py:245-253), per sample.  rec, x: planes [N,3,S,S].
    `rec` itself is also a model output, so its external gradient is added in the same backward."""
    N, Cc, S, _ = rec.shape
    assert norm == "ortho", "frequency reconstruction loss implemented for freq_norm='ortho'"
    cnt_s = Cc * S * S
    Wh = S // 2 + 1
    cnt_f = Cc * S * Wh
    spatial = K.l1_fwd(rec, x, 1.0 / cnt_s)
    d = K.axpby(rec, 1.0, x, -1.0)
    Y = K.dft_rfft2_planes(d.view(N * Cc, S, S))            # zero padding columns contribute |0| = 0
    freq = K.l1_fwd(Y.view(N, -1), None, 1.0 / cnt_f)
    if _needs(tape):
        def bwd():
            gs = tape.pop_grad(spatial)
            gf = tape.pop_grad(freq)
            grec = tape.pop_grad(rec)
            if gs is None and gf is None and grec is None:
                return
            total = None
            if gf is not None:
                dY = K.l1_bwd(Y.view(N, -1), None, gf, 1.0 / cnt_f).view(Y.shape)
                total = K.dft_rfft2_planes_adjoint(dY, S).view(rec.shape)
            if gs is not None:
                total = K.l1_bwd(rec, x, gs, 1.0 / cnt_s, out=total)
            if grec is not None:
                total = grec if total is None else K.axpby(total, 1.0, grec, 1.0)
            tape.grads[id(rec)] = total      # hand the summed gradient to rec's producer
            tape._keep.append(rec)
        tape.record(bwd)
    return spatial, freq


def dynamic_filter(tape, x, proj, diff, w2, b2=None):
    """mask = sigmoid(conv1x1([mean_c proj, max_c proj, diff]) (+ b2)), out = mask * x
    (model/modules.py:94-104, 123-133).  x, proj: [N,h,w,*]; diff: [N,h,w,D] (no grad); w2: [1,2+D,1,1]; b2: [1] — the bias of
    the bias=True variant rides as the weight of one more, constant-1 difference channel (the kernels take any D)."""
    Cx, Cp, D = x.shape[-1], proj.shape[-1], diff.shape[-1]
    x2, p2, d2 = x.view(-1, Cx), proj.view(-1, Cp), diff.view(-1, D)
    w2f = w2.view(-1)
    if b2 is not None:
        d2 = torch.cat([d2, torch.ones_like(d2[:, :1])], 1)
        w2f = torch.cat([w2f.detach(), b2.detach().view(-1)])
    out2, mask, pre, argmax = K.dynfilter_fwd(p2, d2, w2f, x2)
    out = out2.view(x.shape)
    mask4 = mask.view(*x.shape[:-1], 1)
    if _needs(tape):
        def bwd():
            dout = tape.pop_grad(out)
            dmask = tape.pop_grad(mask4)
            if dout is None and dmask is None:
                return
            if dout is None:
                dout = torch.zeros_like(out)
            dx, dlogit, dproj = K.dynfilter_bwd(dout.view(-1, Cx), None if dmask is None else dmask.view(-1),
                                                x2, mask, argmax, w2f, Cp)
            tape.add_grad(x, dx.view(x.shape))
            tape.add_grad(proj, dproj.view(proj.shape))
            # dw2[j] = sum_m dlogit[m] * pre[m][j]   (8 or 5 numbers)
            g = K.gemm_tn(dlogit.view(-1, 1), pre)                 # [1, 2 + D (+ 1)]
            if b2 is None:
                tape.add_param_grad(w2, g.view(w2.shape))
            else:
                tape.add_param_grad(w2, g[:, :-1].reshape(w2.shape))
                tape.add_param_grad(b2, g[:, -1].reshape(b2.shape))
        tape.record(bwd)
    return out, mask4


# ---------------------------------------------------------------------------------------------
# ResNet-variant operators
# ---------------------------------------------------------------------------------------------
def conv_dense_any(tape, x, w, stride, pad, need_dx=True):
    """Dense k x k F.conv2d with symmetric padding and any stride (ResNet convs, model/resnet/exp.py:95-111;
    7x7/2 stem :395; 1x1/2 downsample :235-246).  The data gradient is the transposed-conv gather
    (t = ih + pad - kh, oh = t / stride) with the un-flipped weights."""
    N, Hin, Win, Ci = x.shape
    Co, _, KH, KW = w.shape
    Hout = (Hin + 2 * pad - KH) // stride + 1
    Wout = (Win + 2 * pad - KW) // stride + 1
    g = K.conv_geom(N, Hin, Win, Ci, Hout, Wout, KH, KW, stride, pad, pad, 0)
    wmat = K.weight_layout(w, 0)
    if K.conv_im2col_ok(g, Co, x):
        return _conv_im2col(tape, x, w, wmat, g, need_dx)
    y = K.conv_gather_nt(x, wmat, g)
    if _needs(tape):
        wd = K.weight_layout(w, 2) if need_dx else None

        def bwd():
            dy = tape.pop_grad(y)
            if dy is None:
                return
            if need_dx:
                gd = K.conv_geom(N, Hout, Wout, Co, Hin, Win, KH, KW, stride, pad, pad, 1)
                tape.add_grad(x, K.conv_gather_nt(dy, wd, gd))
            dw = K.conv_gather_wgrad(dy.view(-1, Co), x, g)
            tape.add_param_grad(w, dw.view(Co, KH, KW, Ci).permute(0, 3, 1, 2).contiguous())
        tape.record(bwd)
    return y


def sfconv_dense(tape, x, w, w_freq, alpha, stride, norm, bias=None):
    """SFConv2d.forward (model/resnet/exp.py:36-54): dense 3x3 conv (padding 1; + its bias, :39) + spectral 1x1 branch."""
    spat = bias_add(tape, conv_dense_any(tape, x, w, stride, 1), bias)
    xf = rfft2_cat(tape, x, norm)
    yf = conv1x1(tape, xf, w_freq)
    fr = irfft2_split(tape, yf, norm)
    return sfmix(tape, spat, fr, alpha)


def add_relu(tape, a, b, site=None):
    """`x += shortcut; x = relu(x)` (model/resnet/exp.py:146-147).  site: name under which the parity tests find
    this ReLU's on/off pattern (tape.kinks)."""
    y = K.add_act(a, b, 2)
    if tape is not None and tape.kinks is not None and site is not None:
        tape.kinks[site] = y
    if _needs(tape):
        def bwd():
            dy = tape.pop_grad(y)
            if dy is None:
                return
            g = K.relu_bwd(dy, y)
            tape.add_grad(a, g)
            tape.add_grad(b, g)
        tape.record(bwd)
    return y


def avgpool(tape, x, k):
    """F.adaptive_avg_pool2d to (H/k, W/k) (model/resnet/module_exp.py:30-31)."""
    if k == 1:
        return x
    y = K.avgpool_fwd(x, k)
    if _needs(tape):
        def bwd():
            dy = tape.pop_grad(y)
            if dy is None:
                return
            tape.add_grad(x, K.avgpool_bwd(dy, k))
        tape.record(bwd)
    return y


def maxpool3s2(tape, x, return_arg=False):
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1) (model/resnet/module_exp.py:73-75).
    return_arg: also hand back the uint8 winner map (kh*3+kw per window) for the parity tests."""
    H, W = x.shape[1], x.shape[2]
    y, arg = K.maxpool3s2_fwd(x)
    if _needs(tape):
        def bwd():
            dy = tape.pop_grad(y)
            if dy is None:
                return
            tape.add_grad(x, K.maxpool3s2_bwd(dy, arg, H, W))
        tape.record(bwd)
    return (y, arg) if return_arg else y


def concat_channels(tape, parts):
    """torch.cat(dim=1) in NCHW = channel concat in pixel-major (model/resnet/module_exp.py:32)."""
    y = K.concat_channels(parts)
    if _needs(tape):
        def bwd():
            dy = tape.pop_grad(y)
            if dy is None:
                return
            off = 0
            for p in parts:
                tape.add_grad(p, K.slice_channels(dy, off, p.shape[-1]))
                off += p.shape[-1]
        tape.record(bwd)
    return y


# ---------------------------------------------------------------------------------------------
# Fused MBConv block: deferred BatchNorm, one tape node per block (csrc/fused.hip)
# ---------------------------------------------------------------------------------------------
# bench.py's data-parallel diagnostics: when set to {"bn": [], "ar": []}, every SyncBN sum is bracketed by HIP events on its launch
# stream — (start, end, doubles) — and GradReducer.finish() appends (backward end, last collective waited for, bytes, collectives)
DP_PROFILE = None


class DataParallelCtx:
    """SyncBatchNorm context of the fused path: the fp64 accumulators are summed over the ranks in place (one
    all_reduce between the producing and the consuming kernels), counts are multiplied by the world size."""

    def __init__(self, group=None, exchange=None):
        self.group, self.world, self.synced = group, 1, False
        self.exchange = exchange            # engine.parallel.BnExchange (one kernel per sum) or None (dist.all_reduce)
        if group is not None:
            import torch.distributed as dist
            self.world = dist.get_world_size(group)
            self.synced = self.world > 1 or cfg.force_collectives

    def reduce(self, acc, keep_local=False):
        """Sum `acc` over the ranks in place; returns this rank's own sums (a copy) when keep_local, else None."""
        if not self.synced:
            return None
        via_exchange = self.exchange is not None and acc.numel() <= self.exchange.MAX_DOUBLES
        # the rank's own sums: written by the exchange kernel itself (round 6; a clone launch per BatchNorm backward before)
        loc = (torch.empty_like(acc) if via_exchange else acc.clone()) if keep_local else None
        prof = DP_PROFILE
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if via_exchange:
            self.exchange.allreduce(acc, loc)
        else:
            import torch.distributed as dist
            dist.all_reduce(acc, group=self.group)
        if prof is not None:
            e1.record()
            prof["bn"].append((e0, e1, acc.numel()))
        return loc


class LazyInput:
    """A block input that is still a raw conv output + deferred BatchNorm (the stem in front of block 0):
    backward(dz, sacc) receives the gradient already pushed through the activation and the BatchNorm sums."""

    def __init__(self, bn, backward):
        self.bn, self.backward = bn, backward


def _bn_of(mod, acc, count, act):
    return K.DeferredBN(acc, mod.num_features, count, mod.weight, mod.bias, mod.eps, act,
                        mod.momentum if mod.momentum is not None else 0.1, mod.running_mean, mod.running_var)


def cast(tape, x, dtype):
    """Storage-type boundary of the half-storage trunk (fp16 activations between the fp32 stem / decoder / attention /
    head): y = x.to(dtype), the gradient converted back."""
    if x.dtype == dtype:
        return x
    y = x.to(dtype)
    if _needs(tape):
        def bwd():
            g = tape.pop_grad(y)
            if g is not None:
                tape.add_grad(x, g.reshape(x.shape).to(x.dtype))
        tape.record(bwd)
    return y


def stem_fused(tape, x_pix, w, bn_mod, stride, pad_t, pad_l, Ho, Wo, dp, storage=torch.float32):
    """Stem conv (model/efficientnet/model.py:185-186) whose BatchNorm + swish is left to block 0's depthwise conv.
    storage: dtype the trunk keeps its activations in (torch.float16: the raw conv output is handed on rounded)."""
    h32 = conv_dense(tape, x_pix, w, stride, pad_t, pad_l, Ho, Wo, need_dx=False)
    h = h32 if storage == torch.float32 else h32.to(storage)
    Cc = h.shape[-1]
    M = h.numel() // Cc
    acc = K.zeros64(2 * Cc, h)
    K.colstats(h.view(M, Cc), acc)
    dp.reduce(acc)
    bn = _bn_of(bn_mod, acc, M * dp.world, 1)

    def backward(dz, sacc, is_dz):
        loc = dp.reduce(sacc, keep_local=True)
        dh, dg, db = K.normbwd_apply(h, dz, None, 1.0, bn, is_dz, 1, M, sacc, loc)
        tape.add_param_grad(bn_mod.weight, dg)
        tape.add_param_grad(bn_mod.bias, db)
        tape.add_grad(h32, dh if dh.dtype == h32.dtype else dh.to(h32.dtype))
    return h, LazyInput(bn, backward)


def mbconv_fused(tape, x, blk, keep, keep_prob, wt, dp, lazy_in=None, next_blk=None):
    """MBConvBlock.forward (model/efficientnet/model.py:94-135) in training mode as ONE tape node.
    x [N,H,W,Cin]: the block input (for block 0: the raw stem output, lazy_in its deferred BatchNorm);
    wt: the depthwise weight in tap-major layout [k*k][C]; keep: drop-connect keep vector [N] or None.

    Forward (SF block, stride 1):  expand GEMM -> colstats -> rfft2 of swish(bn0(e)) (also writes the activated
    tensor) -> depthwise conv | spectral GEMM -> irfft2 + SF mix + BN1 statistics -> SE pooling of swish(bn1(d)) ->
    2 small FCs -> BN1 + swish + gate -> project GEMM -> colstats -> BN2 + drop-connect + skip.
    Backward mirrors it with the BatchNorm backward sums accumulated by the kernel that produces the gradient."""
    sp = blk.spec
    N, H, W, Cin = x.shape
    M = N * H * W
    k, stride = sp.k, sp.stride
    pl, pr, pt, pb = sp.pad
    Ho = (H + pt + pb - k) // stride + 1
    Wo = (W + pl + pr - k) // stride + 1
    HWo, Mo = Ho * Wo, N * Ho * Wo
    Ce, Co = sp.cexp, sp.cout
    inv_keep = 1.0 / keep_prob
    dwm = blk._depthwise_conv
    sf = sp.sf_norm is not None
    assert not (sp.skip and lazy_in is not None)

    # ---- expand + BN0 (deferred)
    if sp.expand != 1:
        We = blk._expand_conv.weight.view(Ce, Cin)
        x2 = x.view(M, Cin)
        acc0 = K.zeros64(2 * Ce, x)
        xp = getattr(x, "_ud_planes", None)          # the previous block's residual pass wrote this conv's operand planes itself
        if xp is not None and not (xp.R == M and xp.C == Cin and K.spectral_takes_planes(M, Ce, Cin, We, want_stats=True)):
            xp = None
        (e, done), ectx = K.spectral_fwd(x2 if xp is None else xp, We, stats=acc0, x_absmax=getattr(x, "_ud_absmax", None))          # BN0 statistics in the GEMM epilogue where the launch is plain
        if not done:
            K.colstats(e, acc0)
        e = e.view(N, H, W, Ce)
        dp.reduce(acc0)
        bn0 = _bn_of(blk._bn0, acc0, M * dp.world, 1)
        src, src_bn = e, bn0
    else:
        e = bn0 = None
        src, src_bn = x, (lazy_in.bn if lazy_in is not None else None)

    # ---- depthwise / SF conv -> d (pre-BN1), BN1 statistics
    acc1 = K.zeros64(2 * Ce, x)
    spat = fr = xf = None
    if sf:
        S = H
        s_f, s_i = _fft_scales(S, sp.sf_norm)
        alpha = dwm.sf_coef
        t_fwd, t_wg, t_bwd = _dw_tile_policy(True, k, stride, H, x.dtype == torch.float16)
        t_fused = _dw_bwd_fused_policy(True, k, stride, H, x.dtype == torch.float16)
        Wf = dwm.freq_conv.weight.view(2 * Ce, 2 * Ce)
        if src_bn is not None:
            # a strip kernel needs a = swish(bn0(e)) materialised (rfft2_ex writes it); the tiled ones apply it on load
            bwd_in_fft = K.irfft2_dwbwd_ok(S, k, stride, sp.pad, x.dtype)          # (the backward then needs no materialised a)
            want_a = not (t_fwd and (t_wg or t_fused or bwd_in_fft))
            if K.rfft2_planes_ok(src, src_bn) and K.spectral_takes_planes(N * S * (S // 2 + 1), 2 * Ce, 2 * Ce, Wf):
                # the transform writes the spectral GEMM's fp16 x 2 planes itself (scale from an a-priori bound): no split pass —
                # and, stride 1, the depthwise conv of the plane it holds anyway: no conv kernel either
                if K.rfft2_dw_ok(S, k, stride, sp.pad):
                    xf, a, spat = K.rfft2_ex_planes(src, s_f, 1.0, bn=src_bn, want_act=not (t_wg or t_fused or bwd_in_fft), update=True, dw_wt=wt,
                                                    dw_k=k)
                else:
                    xf, a = K.rfft2_ex_planes(src, s_f, 1.0, bn=src_bn, want_act=want_a, update=True)
            elif K.rfft2_plane_half_ok(src) and K.spectral_takes_plane_half(N * S * (S // 2 + 1), 2 * Ce, 2 * Ce):
                # the mixed-precision mode: the half result laid into the prec-1 plane by the transform (no layout pass)
                if K._P1_DW and K.rfft2_dw_ok(S, k, stride, sp.pad):
                    xf, a, spat = K.rfft2_ex_plane_half(src, s_f, 1.0, bn=src_bn, want_act=not (t_wg or t_fused or bwd_in_fft),
                                                        update=True, dw_wt=wt, dw_k=k)
                else:
                    xf, a = K.rfft2_ex_plane_half(src, s_f, 1.0, bn=src_bn, want_act=want_a, update=True)
            else:
                xf, a = K.rfft2_ex(src, s_f, 1.0, bn=src_bn, want_act=want_a, update=True, want_absmax=True)
        else:
            xf, a = K.rfft2(src, s_f, 1.0, want_absmax=True), src
        if spat is not None:
            pass
        elif t_fwd:
            spat = K.dwtile_fwd(src, wt, k, pt, pl, Ho, Wo, bn=src_bn, stride=stride)
        else:
            spat = K.dwconv_fwd(a, wt, k, stride, pt, pl, Ho, Wo)
        xf_shape = (N, S, S // 2 + 1, 2 * Ce)
        if isinstance(xf, K.Planes):
            yf, sctx = K.spectral_fwd(xf, Wf)
        else:
            yf, sctx = K.spectral_fwd(xf.view(-1, 2 * Ce), Wf, x_absmax=getattr(xf, "_ud_absmax", None))
        yf = yf.view(xf_shape)
        xf = None          # the context holds what the backward needs of it
        if stride == 1:
            d, fr = K.irfft2_mix(yf, s_i, spat, alpha, acc1)          # fr: freq - spat (neither branch is kept)
            spat = None
        else:
            fr = K.irfft2(yf, s_i, 1.0)
            d = K.sfmix_fwd(spat, fr, alpha, True)
            K.colstats(d.view(Mo, Ce), acc1)
        del yf
    else:
        alpha = None
        t_fwd, t_wg, t_bwd = _dw_tile_policy(False, k, stride, H, x.dtype == torch.float16)
        t_fused = _dw_bwd_fused_policy(False, k, stride, H, x.dtype == torch.float16)
        if t_fwd:
            # halo tile staged in LDS with swish(bn0(e)) applied on the way in; BN1 statistics out of the epilogue
            a = None
            d = K.dwtile_fwd(src, wt, k, pt, pl, Ho, Wo, bn=src_bn, stats=acc1, update=True, stride=stride)
        else:
            # the strip kernel (and its weight gradient) re-reads its input per tap: materialise the activation
            a = K.bn_apply(src, src_bn, 1, M, update=True) if src_bn is not None else src
            d = K.dwconv_fwd(a, wt, k, stride, pt, pl, Ho, Wo)
            K.colstats(d.view(Mo, Ce), acc1)
    dp.reduce(acc1)
    bn1 = _bn_of(blk._bn1, acc1, Mo * dp.world, 1)

    # ---- squeeze-excite on swish(bn1(d)), BN1 + swish + gate -> c
    Cs = sp.cse
    wr2, we2 = blk._se_reduce.weight.view(Cs, Ce), blk._se_expand.weight.view(Ce, Cs)
    pool = K.zeros64(N * Ce, x)
    Wp = blk._project_conv.weight.view(Co, Ce)
    # project conv on the planes GEMM: the gated tensor c is written as its planes by se_scale_bn itself, scaled by
    # max |swish(bn1(d))| — which the SE squeeze pass leaves behind (kernels.colsum_bn_amax)
    c_pl = (K._RFFT_PLANES and d.dtype == torch.float32 and Ce % 32 == 0 and cfg.spectral_p2 != "off"
            and K.spectral_takes_planes(Mo, Co, Ce, Wp, want_stats=True))
    if c_pl:
        c_amax = K.colsum_bn_amax(d, bn1, N, HWo, pool, update=True)
    else:
        K.colsum_bn(d, bn1, N, HWo, pool, update=True)
    s1 = K.fc_fwd_d(pool, 1.0 / HWo, wr2, blk._se_reduce.bias, N)
    s2 = K.fc_fwd(s1, we2, blk._se_expand.bias, 1)

    # ---- project + BN2 + drop-connect + skip
    acc2 = K.zeros64(2 * Co, x)
    if not c_pl and K.project_fwd_fused_ok(d, Wp, HWo):
        # thin project conv (the 64 x 64 blocks): gate applied on load, BatchNorm-2 statistics out of the epilogue, c never written
        # (its backward re-makes c as well: csrc/pjbwd.hip)
        p, pctx = K.project_fwd_fused(d, bn1, s2, Wp, N, HWo, stats=acc2)
        done = True
    elif c_pl:
        c = K.se_scale_bn_planes(d, bn1, s2, N, HWo, c_amax)
        (p, done), pctx = K.spectral_fwd(c, Wp, stats=acc2)
    elif (K._P1_DIRECT and d.dtype == torch.float16 and Ce % 8 == 0 and K.spectral_takes_plane_half(Mo, Co, Ce)):
        c = K.se_scale_bn_plane_half(d, bn1, s2, N, HWo)          # the mixed-precision mode: no row-major c, no layout pass
        (p, done), pctx = K.spectral_fwd(c, Wp, stats=acc2)
    else:
        c = K.se_scale_bn(d, bn1, s2, N, HWo, want_absmax=True)
        (p, done), pctx = K.spectral_fwd(c.view(Mo, Ce), Wp, stats=acc2, x_absmax=getattr(c, "_ud_absmax", None))
    if not done:
        K.colstats(p, acc2)
    dp.reduce(acc2)
    bn2 = _bn_of(blk._bn2, acc2, Mo * dp.world, 0)
    p4 = p.view(N, Ho, Wo, Co)
    # the block output is the next block's expand-conv operand: written as that GEMM's planes in the same pass where it takes them
    nxt = None
    if next_blk is not None and next_blk.spec.expand != 1:
        nxt = (Mo, next_blk.spec.cexp, next_blk._expand_conv.weight.view(next_blk.spec.cexp, Co))
    out = K.residual_bn(p4, bn2, keep, inv_keep, x if sp.skip else None, N, HWo, update=True, want_absmax=True, planes_for=nxt)
    if not _needs(tape):
        return out

    def bwd():
        dout = tape.pop_grad(out)
        if dout is None:
            return
        if not (dout.is_contiguous() and dout.shape == out.shape):
            dout = dout.reshape(out.shape).contiguous()
        # ---- BN2 (+ drop-connect scale) backward
        # the project conv's backward on the planes GEMM: the apply pass writes its operand's planes itself, scaled by the bound
        # the sums pass leaves behind (the energy of the incoming gradient: a third sum)
        dp_pl = K.normbwd_planes_ok(p4, pctx)          # (1: fp32, the energy bound as a third sum; 2: half storage, one plane)
        sb2 = K.zeros64((3 if dp_pl == 1 else 2) * Co, x)
        K.normbwd_sums(p4, dout, keep, inv_keep, bn2, False, N, HWo, sb2)
        loc2 = dp.reduce(sb2, keep_local=True)
        if dp_pl:
            pctx.dy, dg2, db2 = K.normbwd_apply_planes(p4, dout, keep, inv_keep, bn2, False, N, HWo, sb2, loc2)
            dp2, dp_amax = (dout if dp_pl == 2 else pctx.w.buf), None          # (device — and, half storage, dtype — for the launch wrappers)
        else:
            dp_, dg2, db2 = K.normbwd_apply(p4, dout, keep, inv_keep, bn2, False, N, HWo, sb2, loc2, want_absmax=True)
            dp2, dp_amax = dp_.view(Mo, Co), getattr(dp_, "_ud_absmax", None)
        tape.add_param_grad(blk._bn2.weight, dg2)
        tape.add_param_grad(blk._bn2.bias, db2)
        # thin project conv (the 64 x 64 blocks): its data gradient dc is never written — each of the two passes over d that need it
        # re-makes its tile from the thin dp (csrc/pjbwd.hip)
        pj_fused = not dp_pl and pctx.plans is None and K.project_bwd_fused_ok(d, Wp, HWo)
        assert pj_fused or pctx.x is not None, "the fused project forward kept no gated tensor: its backward must be the fused one"
        dgate = K.zeros64(N * Ce, x)
        if pj_fused:
            dWp = K.project_bwd_fused_a(d, bn1, s2, dp2, Wp, N, HWo, dgate)
        else:
            dc, dWp = K.spectral_bwd(pctx, dp2, dy_absmax=dp_amax)          # (one launch for both where that fills the chip better)
            dc = dc.view(N, Ho, Wo, Ce)
            K.coldot_bn(dc, d, bn1, N, HWo, dgate)
        tape.add_param_grad(blk._project_conv.weight, dWp)
        # ---- squeeze-excite backward
        dpool, dWe2, dbe2, dWr, dbr = K.se_bwd(dgate, s2, s1, we2, wr2, pool, 1.0 / HWo)
        tape.add_param_grad(blk._se_expand.weight, dWe2)
        tape.add_param_grad(blk._se_expand.bias, dbe2)
        tape.add_param_grad(blk._se_reduce.weight, dWr)
        tape.add_param_grad(blk._se_reduce.bias, dbr)
        # ---- gate + swish + BN1 backward sums in one pass
        sb1 = K.zeros64(2 * Ce, x)
        if pj_fused:
            dz1 = K.project_bwd_fused_b(d, bn1, s2, dpool, 1.0 / HWo, dp2, Wp, N, HWo, sb1)
        else:
            dz1 = K.se_scale_bwd_bn(dc, d, bn1, s2, dpool, 1.0 / HWo, N, HWo, sb1)
        loc1 = dp.reduce(sb1, keep_local=True)
        g_alpha, g_mode, da_f = None, 0, None
        if sf:
            if stride == 1:
                dacc = K.zeros64(64, x)
                dyf_pl = sctx.plans is not None and K.rfft2_planes_ok(d, None)
                en = K.zeros64(Ce, x) if dyf_pl else None
                dd, dg1, db1 = K.normbwd_apply_mix(d, dz1, bn1, N, HWo, sb1, fr, dacc, loc1, energy=en)
                # adjoint of irfft2, x sigmoid(a); the same launch turns the accumulator slots into the gate's gradient
                if dyf_pl:
                    # ... and writes the GEMMs' planes itself: scale from the energy of dd the apply pass just summed
                    dyf, _, dalpha = K.rfft2_ex_planes(dd, s_i, 2.0, gate_alpha=alpha, gate_mode=1, gate_acc=dacc, energy=en)
                elif sctx.plans is not None and K.rfft2_plane_half_ok(dd):
                    dyf, _, dalpha = K.rfft2_ex_plane_half(dd, s_i, 2.0, gate_alpha=alpha, gate_mode=1, gate_acc=dacc)
                else:
                    dyf, _, dalpha = K.rfft2_ex(dd, s_i, 2.0, gate_alpha=alpha, gate_mode=1, gate_acc=dacc, want_absmax=True)
                tape.add_param_grad(alpha, dalpha)
                g_sp, g_alpha, g_mode = dd, alpha, 2                                   # spatial branch: x (1 - sigmoid(a))
            else:
                dd, dg1, db1 = K.normbwd_apply(d, dz1, None, 1.0, bn1, True, N, HWo, sb1, loc1)
                g_sp, dfr, dalpha = K.sfmix_bwd(spat, fr, alpha, dd, True)
                tape.add_param_grad(alpha, dalpha)
                dyf = K.rfft2(dfr, s_i, 2.0, want_absmax=True)
            if isinstance(dyf, K.Planes):
                sctx.dy = dyf                                          # the products' dy operand, already in planes
                # (a tensor of the right device — and, half storage, of the results' dtype — for the launch wrappers)
                dyf2, dyf_amax = (dd if dyf.prec == 1 else sctx.w.buf), None
            else:
                dyf2, dyf_amax = dyf.view(-1, 2 * Ce), getattr(dyf, "_ud_absmax", None)
            dxf, dWf = K.spectral_bwd(sctx, dyf2, dy_absmax=dyf_amax)
            tape.add_param_grad(dwm.freq_conv.weight, dWf)
            dxf = dxf.view(xf_shape)
            # 8 x 8 maps: the adjoint transform's kernel also does the depthwise conv's backward over the planes it holds
            irdw = src_bn is not None and K.irfft2_dwbwd_ok(S, k, stride, sp.pad, x.dtype)
            if not irdw:
                da_f = K.irfft2(dxf, s_f, 0.5)                                         # adjoint of rfft2
        else:
            dd, dg1, db1 = K.normbwd_apply(d, dz1, None, 1.0, bn1, True, N, HWo, sb1, loc1)
            g_sp = dd
        tape.add_param_grad(blk._bn1.weight, dg1)
        tape.add_param_grad(blk._bn1.bias, db1)
        dz0 = dw_f = None
        de_pl = False
        if sf and irdw:
            # (+ the energy of dz0 where the expand conv's backward takes its operand as planes: K.normbwd_apply_planes below)
            de_pl = K.normbwd_planes_ok(src, ectx) if lazy_in is None else 0
            sb0 = K.zeros64((3 if de_pl == 1 else 2) * src.shape[-1], x)
            dz0, dw_f = K.irfft2_dwbwd(dxf, s_f, 0.5, g_sp, src, src_bn, wt, k, g_alpha, g_mode, sb0)
            is_dz = True
            tape.add_param_grad(dwm.weight, dw_f)
        elif t_fused:
            # ---- depthwise data + weight gradient in one pass over (dd, src)
            if src_bn is not None:
                sb0 = K.zeros64(2 * src.shape[-1], x)
                dz0, dw_f = K.dwtile_bwd(g_sp, src, wt, k, pt, pl, bn=src_bn, gate_alpha=g_alpha, gate_mode=g_mode, add=da_f,
                                         sacc=sb0)
                is_dz = True
            else:
                add = da_f
                if sp.skip and add is None:
                    add, skip_done = dout, True
                else:
                    skip_done = not sp.skip
                dx, dw_f = K.dwtile_bwd(g_sp, src, wt, k, pt, pl, gate_alpha=g_alpha, gate_mode=g_mode, add=add)
            tape.add_param_grad(dwm.weight, dw_f)
        elif t_wg:
            tape.add_param_grad(dwm.weight, K.dwtile_bwd_weight(src, g_sp, k, pt, pl, bn=src_bn, gate_alpha=g_alpha,
                                                                gate_mode=g_mode, stride=stride))
        else:
            tape.add_param_grad(dwm.weight, K.dwconv_bwd_weight_ex(a, g_sp, g_alpha, g_mode, k, stride, pt, pl))
        # ---- depthwise data gradient (+ spectral branch), through swish(bn0(.)) when the input is deferred
        if src_bn is not None:
            if dz0 is not None:
                pass                                       # the fused kernel above made dz0 and the BatchNorm sums
            elif t_bwd:
                sb0 = K.zeros64(2 * src.shape[-1], x)
                dz0 = K.dwtile_bwd_data(g_sp, wt, k, pt, pl, H, W, g_alpha, g_mode, da_f, src, src_bn, sb0, stride=stride)
                is_dz = True
            elif stride == 1:
                sb0 = K.zeros64(2 * src.shape[-1], x)
                dz0 = K.dwconv_bwd_data_bn(g_sp, g_alpha, g_mode, wt, da_f, src, src_bn, k, stride, pt, pl, sb0)
                is_dz = True
            else:       # stride 2 (4 of 32 blocks): gather kernel, then the sums as a pass of their own
                sb0 = K.zeros64(2 * src.shape[-1], x)
                dz0 = K.dwconv_bwd_data(g_sp, wt, k, stride, pt, pl, H, W, add=da_f)
                K.normbwd_sums(src, dz0, None, 1.0, src_bn, False, 1, M, sb0)
                is_dz = False
            if lazy_in is not None:
                lazy_in.backward(dz0, sb0, is_dz)
                return
            loc0 = dp.reduce(sb0, keep_local=True)
            if not de_pl and x.dtype == torch.float16 and sp.expand != 1 and K.normbwd_planes_ok(e, ectx) == 2:
                de_pl = 2          # half storage: the plane needs no bound, whichever kernel produced dz0
            if not de_pl and sp.expand != 1 and ectx.plans is None and K.expand_bwd_fused_ok(e, ectx.w, is_dz):
                # thin expand conv (the 128 x 128 / 64 x 64 blocks): BatchNorm backward + both gradients in ONE pass over (dz0, e)
                add = dout.view(M, Cin) if (sp.skip and tape.watch is None and getattr(dout, "_ud_owned", False)) else None
                dx, dWe, dg0, db0 = K.expand_bwd_fused(e, dz0, bn0, sb0, loc0, ectx.x, ectx.w, add=add)
                dx = dx.view(x.shape)
                if sp.skip and add is None:
                    dx = K.axpby(dx, 1.0, dout, 1.0, out=dx)
                tape.add_param_grad(blk._bn0.weight, dg0)
                tape.add_param_grad(blk._bn0.bias, db0)
                tape.add_param_grad(blk._expand_conv.weight, dWe)
                dx._ud_owned = True
                tape.add_grad(x, dx)
                return
            if de_pl:
                ectx.dy, dg0, db0 = K.normbwd_apply_planes(e, dz0, None, 1.0, bn0, is_dz, 1, M, sb0, loc0)
                de2, de_amax = (dz0 if de_pl == 2 else ectx.w.buf), None
            else:
                de, dg0, db0 = K.normbwd_apply(e, dz0, None, 1.0, bn0, is_dz, 1, M, sb0, loc0, want_absmax=True)
                de2, de_amax = de.view(M, Ce), getattr(de, "_ud_absmax", None)
            tape.add_param_grad(blk._bn0.weight, dg0)
            tape.add_param_grad(blk._bn0.bias, db0)
            if sp.skip and tape.watch is None and getattr(dout, "_ud_owned", False):
                dx, dWe = K.spectral_bwd(ectx, de2, out=dout.view(M, Cin), dy_absmax=de_amax)    # + skip gradient
                dx = dx.view(x.shape)
            else:
                dx, dWe = K.spectral_bwd(ectx, de2, dy_absmax=de_amax)
                dx = dx.view(x.shape)
                if sp.skip:
                    dx = K.axpby(dx, 1.0, dout, 1.0, out=dx)
            tape.add_param_grad(blk._expand_conv.weight, dWe)
        else:
            if dw_f is None:
                add = da_f
                if sp.skip and add is None:
                    add, skip_done = dout, True
                else:
                    skip_done = not sp.skip
                if t_bwd:
                    dx = K.dwtile_bwd_data(g_sp, wt, k, pt, pl, H, W, g_alpha, g_mode, add, stride=stride)
                else:
                    dx = K.dwconv_bwd_data_ex(g_sp, g_alpha, g_mode, wt, add, k, stride, pt, pl, H, W)
            if not skip_done:
                dx = K.axpby(dx, 1.0, dout, 1.0, out=dx)
        dx._ud_owned = True
        tape.add_grad(x, dx)
    tape.record(bwd)
    return out
