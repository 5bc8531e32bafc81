"""ud_gemm_p3 (pre-split bf16 planes, LDS-DMA k-loop) against float64 and against the in-kernel split of gemm_x3.hip:
accuracy on every mode, split-K, edge tiles; then graph-replayed timing of both kernels on the spectral shapes.
usage: python tools/check_gemm_p3.py [quick]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidefense_amd import kernels as K

dev = torch.device("cuda:0")
torch.manual_seed(0)
K.CFG.gemm_tune = False


def operands(kind, M, N, Kd):
    """(a, b) fp32 in the layouts of the three products; (a_mode, b_mode)"""
    if kind == "nt":
        return torch.randn(M, Kd, device=dev), torch.randn(N, Kd, device=dev), 0, 0
    if kind == "nn":
        return torch.randn(M, Kd, device=dev), torch.randn(Kd, N, device=dev), 0, 1
    return torch.randn(Kd, M, device=dev), torch.randn(Kd, N, device=dev), 1, 1          # tn


def ref64(kind, a, b):
    a, b = a.double(), b.double()
    return a @ b.t() if kind == "nt" else a @ b if kind == "nn" else a.t() @ b


def x3(kind, a, b, out, M, N, Kd, split=1, cfg=1):
    lda = a.shape[1]
    ldb = b.shape[1]
    am, bm = {"nt": (0, 0), "nn": (0, 1), "tn": (1, 1)}[kind]
    return K._gemm(a, b, out, M, N, Kd, lda, ldb, N, am, bm, 2 if split > 1 else 0, split, cfg=cfg)


bad = 0
shapes = [("nt", 256, 256, 64), ("nt", 128, 128, 32), ("nt", 384, 200, 96), ("nn", 256, 256, 128), ("nn", 300, 264, 160),
          ("tn", 256, 256, 128), ("tn", 192, 320, 256), ("nt", 1280, 3264, 3264), ("nn", 1280, 3264, 3264),
          ("tn", 3264, 3264, 1280), ("nt", 4608, 1344, 1344), ("tn", 672, 672, 17408)]
for kind, M, N, Kd in shapes:
    a, b, am, bm = operands(kind, M, N, Kd)
    ap, bp = K.split_planes(a), K.split_planes(b)
    want = ref64(kind, a, b)
    scale = (a.double().abs() @ b.double().abs().t() if kind == "nt" else a.double().abs() @ b.double().abs() if kind == "nn"
             else a.double().abs().t() @ b.double().abs())
    for split in (1, 2, 3):
        if Kd // 32 < split:
            continue
        out = torch.zeros(M, N, device=dev) if split > 1 else torch.full((M, N), float("nan"), device=dev)
        K._gemm_p3(ap, bp, out, M, N, Kd, am, bm, 2 if split > 1 else 0, split)
        ref = torch.zeros(M, N, device=dev) if split > 1 else torch.empty(M, N, device=dev)
        x3(kind, a, b, ref, M, N, Kd, split)
        torch.cuda.synchronize()
        e = ((out.double() - want).abs() / scale).max().item()
        e3 = ((ref.double() - want).abs() / scale).max().item()
        same = torch.equal(out, ref)
        ok = e < 2e-6 and torch.isfinite(out).all().item()
        bad += not ok
        print(f"{kind} {M}x{N}x{Kd} split {split}: p3 err {e:.2e}  x3 err {e3:.2e}  bitwise-equal-to-x3 {same}  {'ok' if ok else 'FAIL'}",
              flush=True)
    # stream-K: store onto zeros, and add onto an existing term
    out = torch.zeros(M, N, device=dev)
    K._gemm_p3(ap, bp, out, M, N, Kd, am, bm, 0, 1, cfg=0x800)
    base = torch.randn(M, N, device=dev)
    out2 = base.clone()
    K._gemm_p3(ap, bp, out2, M, N, Kd, am, bm, 1, 1, cfg=0x800)
    torch.cuda.synchronize()
    e = ((out.double() - want).abs() / scale).max().item()
    e2 = ((out2.double() - base.double() - want).abs() / (scale + base.double().abs())).max().item()
    ok = e < 2e-6 and e2 < 2e-6 and torch.isfinite(out).all().item()
    bad += not ok
    print(f"   stream-K: err {e:.2e} / onto a term {e2:.2e} {'ok' if ok else 'FAIL'}", flush=True)
    # epilogue statistics
    if am == 0:
        acc = torch.zeros(2 * N, dtype=torch.float64, device=dev)
        out = torch.empty(M, N, device=dev)
        _, done = K._gemm_p3(ap, bp, out, M, N, Kd, am, bm, 0, 1, stats=acc)
        torch.cuda.synchronize()
        s1, s2 = out.double().sum(0), (out.double() ** 2).sum(0)
        es = max(((acc[:N] - s1).abs() / (s1.abs() + 1)).max().item(), ((acc[N:] - s2).abs() / (s2.abs() + 1)).max().item())
        ok = done and es < 1e-9
        bad += not ok
        print(f"   stats: done {done} err {es:.2e} {'ok' if ok else 'FAIL'}", flush=True)
# row sub-range of A (the tail plan)
a, b, am, bm = operands("nt", 640, 256, 128)
ap, bp = K.split_planes(a), K.split_planes(b)
out = torch.empty(256, 256, device=dev)
K._gemm_p3(ap, bp, out, 256, 256, 128, 0, 0, a_row0=384)
torch.cuda.synchronize()
e = (out.double() - a[384:].double() @ b.double().t()).abs().max().item()
print("row offset:", e, "ok" if e < 1e-3 else "FAIL")
bad += e >= 1e-3
# ---- prec 2: two fp16 pieces of the row-scaled operands, three products (nt form; nn runs as nt on the transposed weights)
print("\n# prec 2 (fp16 x 2, row scales) against float64, error relative to sum |a||b| per output; x3 for comparison")
def dist(name, M, Kd):
    g = torch.randn(M, Kd, device=dev)
    if name == "randn": return g
    if name == "same-sign": return g.abs()
    if name == "lognormal3": return g * torch.exp(3 * torch.randn(M, Kd, device=dev))
    if name == "rows-e8": return g * torch.exp(8 * torch.randn(M, 1, device=dev))
    if name == "cols-e4": return g * torch.exp(4 * torch.randn(1, Kd, device=dev))
    if name == "tiny": return g * 1e-30
    if name == "huge": return g * 1e30
    if name == "zero-rows":
        g[::3] = 0
        return g
for name in ("randn", "same-sign", "lognormal3", "rows-e8", "cols-e4", "tiny", "huge", "zero-rows"):
    for M, N, Kd in ((384, 200, 96), (1280, 3264, 3264)):
        a, b = dist(name, M, Kd), dist("randn" if name in ("tiny", "huge") else name, N, Kd)
        want = a.double() @ b.double().t()
        scale = a.double().abs() @ b.double().abs().t() + 1e-300
        ref = torch.empty(M, N, device=dev)
        x3("nt", a, b, ref, M, N, Kd)
        e3 = ((ref.double() - want).abs() / scale).max().item()
        for per_row in (True, False):
            if name == "rows-e8" and not per_row:
                continue          # one scale for a tensor whose rows span 2^+-35: outside the format's range by construction
            ap, bp = K.split_planes(a, prec=2, per_row=per_row), K.split_planes(b, prec=2, per_row=per_row)
            out = torch.full((M, N), float("nan"), device=dev)
            K._gemm_p3(ap, bp, out, M, N, Kd, 0, 0)
            outs = torch.zeros(M, N, device=dev)
            K._gemm_p3(ap, bp, outs, M, N, Kd, 0, 0, 2, 2)
            outk = torch.zeros(M, N, device=dev)
            K._gemm_p3(ap, bp, outk, M, N, Kd, 0, 0, 0, 1, cfg=0x800)
            torch.cuda.synchronize()
            e = ((out.double() - want).abs() / scale).max().item()
            es = ((outs.double() - want).abs() / scale).max().item()
            ek = ((outk.double() - want).abs() / scale).max().item()
            ok = max(e, es, ek) < max(2e-6, 2 * e3) and torch.isfinite(out).all().item()
            bad += not ok
            print(f"{name:11s} {M}x{N}x{Kd} {'row' if per_row else 'tensor'} scales: p2 {e:.2e} split-2 {es:.2e} stream-K {ek:.2e} | x3 {e3:.2e}  {'ok' if ok else 'FAIL'}", flush=True)
# the other two products on tensor-scaled planes: nn (a [M,K] x w [K,N]) and tn (a [K,M]^T x b [K,N])
for kind, M, N, Kd in (("nn", 384, 256, 160), ("nn", 1280, 3264, 3264), ("tn", 192, 320, 256), ("tn", 3264, 3264, 1280), ("tn", 672, 672, 17408)):
    a, b, am, bm = operands(kind, M, N, Kd)
    want = ref64(kind, a, b)
    scale = (a.double().abs() @ b.double().abs() if kind == "nn" else a.double().abs().t() @ b.double().abs())
    ap, bp = K.split_planes(a, prec=2), K.split_planes(b, prec=2)
    res = {}
    for tag, (om, sp, cfg) in {"plain": (0, 1, 0), "split-3": (2, 3, 0), "stream-K": (0, 1, 0x800)}.items():
        if Kd // 32 < sp:
            continue
        out = torch.zeros(M, N, device=dev)
        K._gemm_p3(ap, bp, out, M, N, Kd, am, bm, om, sp, cfg=cfg)
        torch.cuda.synchronize()
        res[tag] = ((out.double() - want).abs() / scale).max().item()
    ref = torch.empty(M, N, device=dev)
    x3(kind, a, b, ref, M, N, Kd)
    e3 = ((ref.double() - want).abs() / scale).max().item()
    ok = max(res.values()) < max(2e-6, 2 * e3)
    bad += not ok
    print(f"{kind} {M}x{N}x{Kd} tensor scales: " + " ".join(f"{k} {v:.2e}" for k, v in res.items()) + f" | x3 {e3:.2e}  {'ok' if ok else 'FAIL'}", flush=True)
print("FAILURES:", bad, flush=True)

if len(sys.argv) > 1 and sys.argv[1] == "quick":
    sys.exit(1 if bad else 0)

print("\n# timing, graph-replayed, us per launch: x3 (best of tiles 128x128 / 128x64 with the listed split) vs p3; split pass of the operands")
rows = [("nt", 4608, 1920, 1920, 2), ("nn", 4608, 1920, 1920, 2), ("nn", 4352, 1920, 1920, 1), ("tn", 1920, 1920, 4608, 2), ("nt", 1280, 3264, 3264, 1),
        ("nn", 1280, 3264, 3264, 1), ("tn", 3264, 3264, 1280, 1), ("nt", 4608, 1344, 1344, 1), ("nn", 4608, 1344, 1344, 1),
        ("tn", 1344, 1344, 4608, 4), ("nt", 17408, 672, 672, 1), ("nn", 17408, 672, 672, 1), ("tn", 672, 672, 17408, 12),
        ("nt", 67584, 384, 384, 1), ("tn", 384, 384, 67584, 48), ("nn", 4096, 4096, 4096, 1), ("nn", 4096, 4096, 8192, 1)]
for kind, M, N, Kd, split in rows:
    a, b, am, bm = operands(kind, M, N, Kd)
    ap, bp = K.split_planes(a), K.split_planes(b)
    out = torch.zeros(M, N, device=dev)
    t3 = min(K._time_launches(lambda: x3(kind, a, b, out, M, N, Kd, split, cfg), n=8) for cfg in (1, 2))
    tp = {}
    for sp in sorted({1, split, 2 * split}):
        if Kd // 32 >= sp:
            tp[sp] = K._time_launches(lambda: K._gemm_p3(ap, bp, out, M, N, Kd, am, bm, 2 if sp > 1 else 0, sp), n=8)
    outz = torch.zeros(M, N, device=dev)
    tsk = K._time_launches(lambda: K._gemm_p3(ap, bp, outz, M, N, Kd, am, bm, 1, 1, cfg=0x800), n=8)
    ts_a = K._time_launches(lambda: K.split_planes(a, ap), n=8)
    ts_b = K._time_launches(lambda: K.split_planes(b, bp), n=8)
    best = min(tp.values())
    fl = 2.0 * M * N * Kd
    ap2, bp2 = K.split_planes(a, prec=2), K.split_planes(b, prec=2)
    tq = {sp: K._time_launches(lambda: K._gemm_p3(ap2, bp2, out, M, N, Kd, am, bm, 2 if sp > 1 else 0, sp), n=8)
          for sp in sorted({1, split, 2 * split}) if Kd // 32 >= sp}
    tq["sk"] = K._time_launches(lambda: K._gemm_p3(ap2, bp2, outz, M, N, Kd, am, bm, 1, 1, cfg=0x800), n=8)
    th_a = K._time_launches(lambda: K.split_planes(a, ap2), n=8)
    th_b = K._time_launches(lambda: K.split_planes(b, bp2), n=8)
    bq = min(tq.values())
    t2 = (" || p2 " + " ".join(f"{k} {t * 1e3:6.1f}" for k, t in tq.items()) +
          f" best {fl / bq / 1e9:5.1f} TF x{t3 / bq:.2f} absmax+split A {th_a * 1e3:.1f} B {th_b * 1e3:.1f}")
    print(f"{kind} {M}x{N}x{Kd}: x3 {t3 * 1e3:7.1f} ({fl / t3 / 1e9:5.1f} TF)  p3 " +
          " ".join(f"s{sp} {t * 1e3:7.1f}" for sp, t in tp.items()) +
          f"  best {fl / best / 1e9:5.1f} TF  x{t3 / best:.2f} | stream-K {tsk * 1e3:7.1f} ({fl / tsk / 1e9:5.1f} TF) x{t3 / tsk:.2f} | split A {ts_a * 1e3:.1f} B {ts_b * 1e3:.1f}" + t2, flush=True)
