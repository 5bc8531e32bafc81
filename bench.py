"""bench.py — images/sec of ONE forward+backward of UniDefenseModelEb4 (pass-1 loss of the train step,
engine/abstract_engine.py:210-281 in the reference) at 256x256, bs=32 per GPU, fp32, synthetic inputs.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (metric/unit from BASELINE.json; value = whole-job images/s with inputs
resident in HBM; weak scaling: bs=32 per GPU).  Extra objects:
  roofline     — dominant kernels = the matrix-pipe GEMM family: gemm_p3_kernel<prec 2> (operands pre-split into two fp16
                 planes, THREE fp16 MFMAs per fp32 product tile) + gemm_x3_kernel (in-kernel bf16 x 3 split, SIX MFMAs):
                 algorithmic FLOPs (2MNK) of their launches / their HIP-event durations (3 eager instrumented steps after
                 the timed region); peak = the pipe's dense 16-bit peak (2.5 PFLOP/s) / executed MFMAs per algorithmic
                 product, so frac = executed MFMA work / pipe peak (`frac_six_product_equiv`: the same work priced at six
                 MFMAs per product, the round-3 definition); `hbm`: the same launches' algorithmic bytes against 8 TB/s;
                 `traffic`: PMC-measured HBM bytes per launch (profiles/).
  extra        — bounded child-process measurements of the other BASELINE configs ([4] f16 bs 64, [3] UDR50 320^2 bs 16,
                 [0] UDR18 128^2 bs 8), the engine's two-pass train step and UDEB4 at 380^2 (N = 1 only).
  cpu_baseline — the oracle (CPU restatement, "port") timed on the host cores (CPU quota of the job) on a bounded
                 sample (bs 8, ~10 s), in a child process after the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

MFMA_F32_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
BF16_PEAK_TFLOPS = 2500.0         # same guide: "Peak BF16/FP16 MFMA ~2.5 PF dense"
X3_MFMA_PER_PRODUCT = 6           # gemm_x3.hip: six bf16 piece products per fp32 product
LAMBDAS = dict(lambda_triplet=0.1, lambda_recons=0.1, lambda_freq=1.0, lambda_mask=0.1)   # uniatt/Prot1/model_udeb4.yml


FUSED_LOSS_TAIL = os.environ.get("UD_BENCH_FUSED_TAIL", "1") == "1"          # A/B: the loss tail as torch ops


def pass1_loss(out, tgt, n_real, losses):
    """the engine's pass-1 loss (AbstractEngine._pass1; reference engine/abstract_engine.py:232-270): the same scalar tail the
    engine runs — two HIP launches (loss/pass_tail.py) where it applies, the torch formulation otherwise"""
    from unidefense_amd.loss.pass_tail import pass_tail
    f = pass_tail(out, tgt, n_real, tgt.shape[0] - n_real, {"softmax": losses["cross_entropy"], "triplet": losses["aw_triplet"]},
                  dict(cls=1.0, mask=LAMBDAS["lambda_mask"], triplet=LAMBDAS["lambda_triplet"], rec=LAMBDAS["lambda_recons"],
                       freq=LAMBDAS["lambda_freq"])) if FUSED_LOSS_TAIL else None
    if f is not None:
        return f["total"]
    ld = out["loss_dict"]
    trip = sum(losses["aw_triplet"](f, tgt) for f in ld["triplet"])
    cls = losses["cross_entropy"](out["cls_out"], tgt)
    return cls + LAMBDAS["lambda_mask"] * (ld["freq_mask"].mean() + ld["spat_mask"].mean()) + \
        LAMBDAS["lambda_triplet"] * trip + LAMBDAS["lambda_recons"] * ld["spatial"].narrow(0, 0, n_real).mean() + \
        LAMBDAS["lambda_freq"] * ld["freq"].narrow(0, 0, n_real).mean()


def host_cores():
    """CPU cores this process may actually use: min(affinity mask, cgroup v2 cpu.max quota).  The GPU boxes show
    256 logical CPUs but run the job under a 16-CPU quota — 256 OpenMP threads there are ~20x SLOWER than 16."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline_measure(bs=8, warm=2, iters=5):
    """Oracle forward + pass-1 loss + backward on the host cores, bounded sample (runs in a child process).
    SURVEY.md §8(d) protocol: 2 warm-up + 5 timed iterations, median, all cores of the job; plus a 1-thread figure on a
    smaller sample (bs 2, 1 + 2 iterations) so that the whole measurement stays at ~25 s of CPU work."""
    from oracle import param_fill
    from tests import oracle_util as ou

    def run(threads, bs_, warm_, iters_):
        torch.set_num_threads(threads)
        x = param_fill.make_input(bs_, 256, seed=0)
        tgt = param_fill.make_labels(bs_)
        rng = ou.make_rng(bs_, 1, 0.5)
        sd = ou.oracle_state(-10.0, 0.0, requires_grad=True)
        ts = []
        for it in range(warm_ + iters_):
            for v in sd.values():
                v.grad = None
            t0 = time.perf_counter()
            ou.oracle_train_pass1(sd, x, tgt, rng, 0.5)
            if it >= warm_:
                ts.append(time.perf_counter() - t0)
        ts.sort()
        return bs_ / ts[len(ts) // 2]
    cores = host_cores()
    v = run(cores, bs, warm, iters)
    v1 = run(1, 2, 1, 2)
    return {"value": v, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"oracle (pure-torch CPU restatement of the reference) fwd + pass-1 loss + bwd, UDEB4 256x256 "
                      f"bs={bs}, fp32, median of {iters} after {warm} warm-ups, torch threads={cores} "
                      f"(= CPU quota of the job; the host shows {os.cpu_count()} logical CPUs)",
            "value_1thread": v1, "sample_1thread": "same step, bs=2, 1 torch thread, median of 2 after 1 warm-up"}


def cpu_baseline(timeout_s=300):
    """Run the measurement in a child process (own thread pool, no GPU context) under a wall-clock bound so that a
    mis-sized host can never stall the bench line."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"], capture_output=True,
                           text=True, timeout=timeout_s, cwd=ROOT)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode == 0 and lines:
            return json.loads(lines[-1])
        return {"value": None, "unit": "images/sec", "cores": host_cores(), "kind": "port",
                "sample": f"failed (exit {r.returncode}): {r.stderr[-200:]}"}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "images/sec", "cores": host_cores(), "kind": "port",
                "sample": f"not finished within {timeout_s} s"}


def train_step_measure(bs):
    """the engine's full two-pass train step (AbstractEngine.train_unidefense_model, abstract_engine.py:207-381: clean pass +
    perturbed pass, two AdamW steps) on UDEB4 256x256 — informational, run in a child process of the default bench"""
    from unidefense_amd.engine import AbstractEngine
    from unidefense_amd.engine.optim import build_optimizer
    from unidefense_amd.loss import LOSSES
    from unidefense_amd.model import load_model
    import contextlib
    dev = torch.device("cuda:0")
    with contextlib.redirect_stdout(sys.stderr):
        m = load_model("UDEB4")(extractor="efficientnet-b4", num_classes=2, drop_rate=0.5).to(dev).train()
    eng = AbstractEngine({"config": dict(lambda_triplet=0.1, lambda_recons=0.1, lambda_freq=1.0, lambda_mask=0.1, lambda_fac=0.1)})
    eng.model, eng.device, eng.num_steps, eng.warmup_step = m, dev, 1000, 0
    eng.optimizer = build_optimizer(m, dict(name="adamw", lr=1e-4, betas=[0.9, 0.999], weight_decay=5e-6, amsgrad=True))
    eng.scheduler = torch.optim.lr_scheduler.StepLR(eng.optimizer, step_size=22500, gamma=0.5)
    eng.loss_criterion = {"softmax": LOSSES["cross_entropy"], "triplet": LOSSES["aw_triplet"], "kl_div": LOSSES["kl_div"],
                          "fac": LOSSES["factorization"]}
    x = (2 * torch.rand(bs, 3, 256, 256) - 1).to(dev)
    tgt = torch.tensor([0] * (bs // 2) + [1] * (bs // 2), device=dev)
    scaler = torch.amp.GradScaler("cuda", init_scale=2 ** 10, enabled=False)

    def step(i):
        eng.optimizer.zero_grad()
        return eng.train_unidefense_model(x, tgt, 200 + i, scaler, bs // 2, bs // 2)
    for i in range(3):
        step(i)
    # The perturbation of pass 2 is drawn on the host per step (six kinds of very different cost; the style branch under
    # preserve_color runs CORAL, whose 3 x 3 SVD is a host LAPACK call behind two device read-backs): the same seeded sequence
    # of branches in every round and run (that of tools/bench_train_step.py), THREE rounds of 8 steps, the median round
    # reported; steps are timed one by one (a synchronize each) so that the CORAL steps can be told apart.
    from unidefense_amd.model import perturb as _pert
    took_coral = [False]
    coral0 = _pert.coral

    def coral_flagged(*a, **k):
        took_coral[0] = True
        return coral0(*a, **k)
    _pert.coral = coral_flagged
    n, rounds, per_step = 8, [], []
    for _ in range(3):
        torch.manual_seed(0)
        torch.cuda.synchronize()
        t_round = 0.0
        for i in range(n):
            took_coral[0] = False
            t0 = time.perf_counter()
            step(i)
            torch.cuda.synchronize()
            dt_i = time.perf_counter() - t0
            t_round += dt_i
            per_step.append((took_coral[0], dt_i))
        rounds.append(t_round / n)
    _pert.coral = coral0
    rounds.sort()
    dt = rounds[1]
    coral = [t for c, t in per_step if c]
    plain = [t for c, t in per_step if not c]
    return {"what": f"engine two-pass train step (2 x fwd+bwd + 2 AdamW), UDEB4 256x256 bs {bs}", "ms_per_step": 1e3 * dt,
            "value": bs / dt, "unit": "train images/sec (each image goes through 2 passes)", "steps": n,
            "rounds_ms": [1e3 * r for r in rounds], "protocol": "median of 3 rounds x 8 steps, the same seeded branch sequence",
            "ms_per_step_coral_branch": 1e3 * sum(coral) / len(coral) if coral else None, "coral_steps": len(coral),
            "ms_per_step_other_branches": 1e3 * sum(plain) / len(plain) if plain else None}


def extra_measurements(budget_s=100):
    """Driver-visible numbers of the other BASELINE configs: bounded 5-step runs in CHILD processes after the timed region
    (configs[4] f16 bs 64, configs[3] UDR50 320^2 bs 16, configs[0] UDR18 128^2 bs 8, the two-pass train step, UDEB4 at 380^2)."""
    import subprocess
    me = os.path.abspath(__file__)
    common = ["--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-extra"]
    jobs = [("configs[4]: UDEB4 256x256 fp16 MFMA + half storage, bs 64", ["--dtype", "f16", "--batch", "64"] + common),
            ("configs[3]: UDR50 320x320 bs 16", ["--model", "UDR50", "--size", "320", "--batch", "16"] + common),
            ("configs[0]: UDR18 128x128 bs 8", ["--model", "UDR18", "--size", "128", "--batch", "8"] + common),
            ("two-pass train step", ["--train-step"]),
            ("UDEB4 at the reference YAMLs' 380x380, bs 32", ["--size", "380"] + common),
            ("UDEB4 at 380x380 with the reference YAMLs' own batch, 10 real + 10 fake per GPU", ["--size", "380", "--batch", "20"] + common)]
    out, t_start = [], time.perf_counter()
    for name, argv in jobs:
        left = budget_s - (time.perf_counter() - t_start)
        if left < 8:
            out.append({"what": name, "value": None, "note": "skipped: the extras' time budget was spent"})
            continue
        try:
            r = subprocess.run([sys.executable, me] + argv, capture_output=True, text=True, timeout=left, cwd=ROOT)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            d = json.loads(lines[-1]) if (r.returncode == 0 and lines) else None
        except (subprocess.TimeoutExpired, ValueError):
            d = None
        if d is None:
            out.append({"what": name, "value": None, "note": "failed or not finished in time"})
        elif "metric" in d:
            rf = d.get("roofline", {})
            out.append({"what": name, "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
                        "dtype": d["dtype"], "exec": d["config"].get("exec"),
                        "step_achieved_hbm_frac": rf.get("step_achieved_hbm_frac")})
        else:
            out.append(d)
    return out


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start `python -m torch.distributed.run --nproc-per-node N` over
    this same command line as a CHILD process (never exec; nothing in this process has touched the GPU yet), relay its
    output (rank 0 prints the JSON line) and return its exit code.  Fails with a clear message when the box has fewer
    than N GPUs (torch.cuda.device_count() does not initialise the GPU)."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n:
        print(f"bench.py: --gpus {n} requested but this box exposes {have} GPU(s); nothing was run", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this pool (RCCL across processes)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env, cwd=ROOT).returncode


def _gemm_src_sha():
    """sha256 over the GEMM kernels' sources (csrc/gemm*.hip / .h): what a PMC summary must have been measured on"""
    import glob
    import hashlib
    root = os.path.dirname(os.path.abspath(__file__))
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "unidefense_amd", "csrc", "gemm*"))):
        if f.endswith((".hip", ".h")):
            with open(f, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def _pmc_traffic(args, bs, family=None):
    """roofline.traffic: HBM bytes per GEMM launch from the committed PMC summary of this same workload
    (tools/gpu_traffic.sh: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 half-count correction).  PMC passes
    cannot run inside the timed bench, so the number is read from profiles/ — and only from a summary stamped with the
    hash of the GEMM sources this tree builds (`gemm_src_sha`): a summary of other kernels is not these kernels' traffic.
    family: a key of the summary's `families` (one kernel), None: the matrix-pipe family.  (None, None) when nothing matches."""
    if (args.model, args.size, bs) != ("UDEB4", 256, 32):
        return None, None
    import glob
    root = os.path.dirname(os.path.abspath(__file__))
    sha = _gemm_src_sha()
    for f in sorted(glob.glob(os.path.join(root, "profiles", "r*", "hbm_traffic_gemm.json")), reverse=True):
        try:
            with open(f) as fh:
                d = json.load(fh)
            if d.get("gemm_src_sha") != sha:
                continue
            fam = d.get("families", {}).get(family) if family else d.get("gemm_family")
            if fam is None:
                continue
            return float(fam["hbm_bytes_per_launch"]), os.path.relpath(f, root)
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def _csrc_sha():
    import glob
    import hashlib
    root = os.path.dirname(os.path.abspath(__file__))
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "unidefense_amd", "csrc", "*"))):
        if f.endswith((".hip", ".h")):
            with open(f, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def _committed(args, bs, name, sha_key, sha):
    """a committed summary under profiles/rNN/ measured on THIS tree's kernels (its sha stamp must match), newest first"""
    if (args.model, args.size, bs, args.dtype) != ("UDEB4", 256, 32, "f32"):
        return None
    import glob
    root = os.path.dirname(os.path.abspath(__file__))
    for f in sorted(glob.glob(os.path.join(root, "profiles", "r*", name)), reverse=True):
        try:
            with open(f) as fh:
                d = json.load(fh)
            if d.get(sha_key) == sha:
                d["file"] = os.path.relpath(f, root)
                return d
        except (OSError, ValueError):
            continue
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--model", default="UDEB4", choices=["UDEB4", "UDR18", "UDR50"],
                    help="informational runs of the other BASELINE configs (the contract line is UDEB4)")
    ap.add_argument("--size", type=int, default=256, help="input resolution (UDR18: 128, UDR50: 256 or 320)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f16"],
                    help="f32: the contract line (fp32-accurate GEMMs).  f16: informational, BASELINE configs[4]: fp16 MFMA "
                         "operands + fp32 accumulation in every plain GEMM (ud_gemm path 3); use --batch 64")
    ap.add_argument("--storage", default="f16", choices=["f16", "f32"],
                    help="with --dtype f16: activation storage of the MBConv trunk (f16: half storage, the full configs[4] mode)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="do not capture the step into a hipGraph")
    ap.add_argument("--gemm-table", default=None, help="write a per-shape GEMM timing table to this file")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the bounded secondary measurements (the other BASELINE configs, the engine's two-pass train step)")
    ap.add_argument("--train-step", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline_measure()), flush=True)
        return
    if args.train_step:
        print(json.dumps(train_step_measure(args.batch)), flush=True)
        return

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from unidefense_amd.config import cfg
    force = cfg.force_collectives     # UD_FORCE_COLLECTIVES=1: 1-GPU exercise of the RCCL path (tape.py)
    if world > 1 or force:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    from unidefense_amd import kernels as K
    from unidefense_amd.loss import LOSSES
    from unidefense_amd.model import load_model
    from unidefense_amd.engine.parallel import wrap_data_parallel

    if args.dtype == "f16":
        from unidefense_amd import lib as _udlib
        _udlib.call("ud_gemm_set_path", 3)
    torch.manual_seed(1234)
    ctor = dict(extractor="efficientnet-b4") if args.model == "UDEB4" else {}
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):       # the loader announces the model like the reference's does: keep
        model = load_model(args.model)(num_classes=2, drop_rate=0.5, **ctor).to(dev).train()     # stdout to the ONE JSON line
    if args.dtype == "f16" and args.storage == "f16":
        model.half_storage = True            # MBConv trunk activations / activation gradients in fp16 (configs[4])
    model = wrap_data_parallel(model, local_rank) if (world > 1 or force) else model
    bs = args.batch
    g = torch.Generator().manual_seed(100 + rank)
    x = (2.0 * torch.rand(bs, 3, args.size, args.size, generator=g) - 1.0).to(dev)
    tgt = torch.tensor([0] * (bs // 2) + [1] * (bs // 2), device=dev)
    for k_ in ("aw_triplet",):
        LOSSES[k_].n_real = bs // 2
    params = [p for p in model.parameters() if p.requires_grad]

    def step():
        for p in params:
            p.grad = None
        out = model(x)
        loss = pass1_loss(out, tgt, bs // 2, LOSSES)
        # f16: the engine's GradScaler scale (forgery_engine.py:228), so that half gradients do not underflow
        (loss * 1024.0 if args.dtype == "f16" else loss).backward()
        return loss

    # ---- execution mode: the whole step (zero grads, forward, loss, backward [, gradient exchange]) is
    # captured once into a hipGraph and replayed — ~2000 kernel launches per step with no host work between
    # them.  Any capture failure falls back to eager launches (reported in "exec").
    exec_mode = "eager"
    run = step
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(max(1, min(args.warmup, 2))):      # eager warm-up (also sets kernel attributes, caches)
            loss = step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    if not args.eager:
        try:
            graph = torch.cuda.CUDAGraph()
            # With a process group alive, its watchdog thread polls hipEventQuery on the work items of earlier
            # collectives; under the default "global" capture mode that call is illegal while ANY thread captures
            # and takes the process down.  thread_local restricts the check to this thread; draining the
            # outstanding work first keeps the watchdog's list empty during the capture anyway.
            mode = "thread_local" if dist.is_initialized() else "global"
            if dist.is_initialized():
                dist.barrier()
                torch.cuda.synchronize()
                time.sleep(1.0)
            with torch.cuda.graph(graph, capture_error_mode=mode):
                static_loss = step()
            torch.cuda.synchronize()

            def run():
                graph.replay()
                return static_loss
            exec_mode = "hipgraph"
        except Exception as e:      # noqa: BLE001 — keep the bench alive, say what happened
            print(f"[bench] graph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            torch.cuda.synchronize()
            run = step
        if world > 1:
            # all ranks must issue the same sequence of collectives: if the capture failed anywhere, everyone runs eagerly
            ok = torch.tensor([1.0 if exec_mode == "hipgraph" else 0.0], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if ok.item() < 1.0 and exec_mode == "hipgraph":
                print("[bench] another rank could not capture the step; running eagerly", file=sys.stderr)
                run, exec_mode = step, "eager"
    for _ in range(args.warmup):
        loss = run()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = run()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    bn_exchange = getattr(model, "bn_exchange", None)
    if bn_exchange is not None and bn_exchange.ok:
        bn_exchange.check()          # a rank that timed out waiting for a peer's SyncBN sums: fail loudly, no line
    n_ranks_seen = 1
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    if dist.is_initialized():
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                                  # every rank that really took part adds 1 (over RCCL)
        n_ranks_seen = int(ones.item())
    # ---- roofline of the dominant kernel: the same step, eager, every ud_gemm launch bracketed by HIP events
    # on its launch stream (events cannot be read back from inside a replayed graph)
    prof_steps = min(args.steps, 3)
    from unidefense_amd import tape as T
    K.GEMM_PROFILE = []
    if dist.is_initialized():
        T.DP_PROFILE = {"bn": [], "ar": []}
    for _ in range(prof_steps):
        step()
    torch.cuda.synchronize()
    prof, K.GEMM_PROFILE = K.GEMM_PROFILE, None
    # ---- data-parallel diagnostics (N > 1, or UD_FORCE_COLLECTIVES=1 on one GPU): the same eager instrumented steps.
    # syncbn_exchange_ms: device time of the ~210 SyncBN sums of a step (HIP events around each on its stream: at N > 1 this
    # INCLUDES the wait for the slowest peer, i.e. rank skew + xGMI latency on the forward's / backward's dependency chain);
    # allreduce_exposed_ms: end of the backward's own kernels -> last gradient collective waited for (what the overlap did
    # not hide); both the MAX over the ranks.
    dp_diag = None
    if T.DP_PROFILE is not None:
        dpp, T.DP_PROFILE = T.DP_PROFILE, None
        bn_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in dpp["bn"]) / prof_steps
        ar_ms = sum(e0.elapsed_time(e1) for e0, e1, _, _ in dpp["ar"]) / prof_steps
        worst = torch.tensor([bn_ms, ar_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(worst, op=dist.ReduceOp.MAX)
        dp_diag = {"syncbn_exchanges_per_step": len(dpp["bn"]) / prof_steps,
                   "syncbn_exchange_ms": float(worst[0]), "syncbn_exchange_ms_rank0": bn_ms,
                   "syncbn_doubles_per_step": sum(n for _, _, n in dpp["bn"]) / prof_steps,
                   "allreduce_exposed_ms": float(worst[1]), "allreduce_exposed_ms_rank0": ar_ms,
                   "allreduce_bytes": sum(b for _, _, b, _ in dpp["ar"]) / prof_steps,
                   "allreduce_collectives": sum(c for _, _, _, c in dpp["ar"]) / prof_steps,
                   "measured": f"{prof_steps} eager instrumented steps after the timed region, HIP events on the launch stream; "
                               "MAX over the ranks"}

    if rank == 0 and args.gemm_table:
        agg = {}
        for e0, e1, f, key, path, _nb in prof:
            a = agg.setdefault(key + (path,), [0, 0.0, 0.0])
            a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += f
        rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
        with open(args.gemm_table, "w") as fh:
            fh.write("# per-shape GEMM timing, %d eager instrumented steps; pipe: 4 = the planes kernel gemm_p3_kernel (f32 mode: prec 2, pre-split fp16 x 2 "
                     "planes, 3 MFMAs per product; --dtype f16: prec 1, one fp16 plane, 1 MFMA), 2 = gemm_x3_kernel (in-kernel bf16 x 3 "
                     "split, 6 MFMAs), 3 = gemm_x3_kernel's fp16 form (--dtype f16), 1 = gemm_kernel (fp32 pipe)\n" % prof_steps)
            fh.write("M N K a_mode b_mode split batch pipe | calls ms_total ms_per_step TFLOP/s gflop_per_step\n")
            for key, (cnt, ms, fl) in rows:
                fh.write("%7d %5d %6d %d %d %3d %3d %d | %4d %8.3f %8.3f %7.1f %9.3f\n" %
                         (*key, cnt, ms, ms / prof_steps, fl / (ms * 1e-3) / 1e12 if ms > 0 else 0, fl / 1e9 / prof_steps))
    if rank == 0:
        # GEMM families on the matrix pipe: gemm_p3_kernel prec 2 (path 4: pre-split fp16 x 2 planes, THREE fp16 MFMAs per
        # product tile), gemm_x3_kernel (path 2: in-kernel bf16 x 3 split, SIX bf16 MFMAs; path 3: its one-MFMA fp16 mode);
        # everything else is gemm_kernel on the fp32 pipe.  The roofline's kernel is the family with the most time.
        p2 = [p for p in prof if p[4] == 4]
        x3 = [p for p in prof if p[4] in (2, 3)]
        f32 = [p for p in prof if p[4] not in (2, 3, 4)]

        def fam(ps, mfma_per_product):
            ms = sum(p[0].elapsed_time(p[1]) for p in ps)
            fl = sum(p[2] for p in ps)
            ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            by = sum(p[5] for p in ps)
            return {"launches_per_step": len(ps) / prof_steps, "ms_per_step": ms / prof_steps, "gflop_per_step": fl / 1e9 / prof_steps,
                    "achieved_tflops_fp32_equiv": ach, "mfma_per_product": mfma_per_product,
                    "executed_mfma_tflops": ach * mfma_per_product, "frac_of_pipe": ach * mfma_per_product / BF16_PEAK_TFLOPS,
                    "algorithmic_bytes_per_launch": by / max(len(ps), 1),
                    "hbm_gbs": by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0}
        fam_p2 = fam(p2, 3)
        fam_x3 = fam(x3, 1 if args.dtype == "f16" else X3_MFMA_PER_PRODUCT)
        # the roofline's kernel: BOTH matrix-pipe kernels as one family (they share the pipe, and which of the two has more time
        # flips with the host pacing of the eager instrumented steps): achieved = their algorithmic fp32 FLOPs / their time,
        # executed MFMA work = 3 x the planes kernel's + 6 x the in-kernel-split kernel's algorithmic FLOPs
        dom_name = "matrix-pipe GEMM family: gemm_p3_kernel<prec 2> + gemm_x3_kernel"
        pipe_ms = fam_p2["ms_per_step"] + fam_x3["ms_per_step"]
        pipe_gf = fam_p2["gflop_per_step"] + fam_x3["gflop_per_step"]
        exec_gf = fam_p2["gflop_per_step"] * fam_p2["mfma_per_product"] + fam_x3["gflop_per_step"] * fam_x3["mfma_per_product"]
        n_pipe = max(len(p2) + len(x3), 1)
        dom = {"achieved_tflops_fp32_equiv": pipe_gf / pipe_ms if pipe_ms > 0 else 0.0,
               "mfma_per_product": exec_gf / pipe_gf if pipe_gf > 0 else 6.0,
               "algorithmic_bytes_per_launch": (sum(p[5] for p in p2) + sum(p[5] for p in x3)) / n_pipe,
               "hbm_gbs": (sum(p[5] for p in p2) + sum(p[5] for p in x3)) / prof_steps / (pipe_ms * 1e-3) / 1e9 if pipe_ms > 0 else 0.0,
               "launches_per_step": n_pipe / prof_steps}
        mfma_per_product = dom["mfma_per_product"]
        x3_ms = sum(p[0].elapsed_time(p[1]) for p in x3)
        x3_flops = sum(p[2] for p in x3)
        f32_ms = sum(p[0].elapsed_time(p[1]) for p in f32)
        f32_flops = sum(p[2] for p in f32)
        gemm_ms = x3_ms + f32_ms + fam_p2["ms_per_step"] * prof_steps
        gemm_flops = x3_flops + f32_flops + fam_p2["gflop_per_step"] * 1e9 * prof_steps
        achieved = dom["achieved_tflops_fp32_equiv"]                                   # algorithmic fp32 TFLOP/s of the dominant family
        achieved_all = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        alg_bytes = dom["algorithmic_bytes_per_launch"]
        hbm_gbs = dom["hbm_gbs"]
        n_dom = dom["launches_per_step"]
        traffic, traffic_src = _pmc_traffic(args, bs)
        line = {
            "metric": ("images/sec fwd+bwd (256x256, EffNet-b4)" if (args.model, args.size) == ("UDEB4", 256)
                       else f"images/sec fwd+bwd ({args.size}x{args.size}, {args.model})")
            + ((" [informational: fp16 MFMA operands, fp32 accumulate, %s trunk storage]" % args.storage) if args.dtype == "f16" else ""),
            "value": world * bs * args.steps / elapsed,
            "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": ("UDEB4 (EfficientNet-b4 + SFConv) 256x256 fwd + pass-1 loss + bwd, "
                                    "spatial+frequency branches on, bs=32/GPU (BASELINE configs[1]/[2])")
                       if (args.model, args.size, bs, args.dtype) == ("UDEB4", 256, 32, "f32") else
                       f"{args.model} {args.size}x{args.size} fwd + pass-1 loss + bwd, bs={bs}/GPU, dtype {args.dtype} (informational)",
                       "global_batch": world * bs, "parallelism": f"dp{world}", "exec": exec_mode,
                       "n_ranks_seen": n_ranks_seen,
                       # SyncBN sums of the fused path: "peer-exchange" (csrc/xchg.hip mailboxes over xGMI) or "all_reduce" (RCCL)
                       "syncbn": ("peer-exchange" if (bn_exchange is not None and bn_exchange.ok) else "all_reduce")
                       if (world > 1 or force) else "local",
                       "final_loss": float(loss.detach()),
                       # checksum of the step's result (tests: the data-parallel path at world size 1 must give
                       # the plain step's gradients)
                       "grad_l1": float(sum(p.grad.double().abs().sum() for p in params if p.grad is not None)),
                       "deterministic": cfg.deterministic},
            "roofline": {"bound": "hbm" if args.dtype == "f16" else "mfma",
                         "kernel": dom_name + ".  gemm_p3_kernel<prec 2> "
                                   "(csrc/gemm_p3.hip): fp32-accurate GEMM from operands pre-split into two fp16 pieces (one "
                                   "power-of-two scale per tensor), LDS-DMA loader waves + MFMA waves, THREE "
                                   "v_mfma_f32_32x32x16_f16 per fp32 product tile; gemm_x3_kernel (csrc/gemm_x3.hip): every fp32 "
                                   "operand split in the k-loop into 3 bf16 pieces, SIX v_mfma_f32_32x32x16_bf16 per tile.  "
                                   "achieved = algorithmic fp32 FLOPs (2MNK) of the family's launches / their HIP-event time; "
                                   "peak = the pipe's dense 16-bit peak (2500 TFLOP/s) / executed MFMAs per algorithmic product, "
                                   "so frac = executed MFMA work / pipe peak",
                         "achieved": achieved, "peak": BF16_PEAK_TFLOPS / mfma_per_product, "unit": "TFLOP/s",
                         "frac": achieved * mfma_per_product / BF16_PEAK_TFLOPS,
                         "executed_mfma_tflops": achieved * mfma_per_product, "pipe_peak": BF16_PEAK_TFLOPS,
                         "mfma_per_product": mfma_per_product,
                         "frac_of_pipe": achieved * mfma_per_product / BF16_PEAK_TFLOPS,
                         # for comparison with rounds 1-3, whose GEMMs all executed six MFMAs per product: the same algorithmic
                         # throughput priced at six (what the round-3 kernel would have had to execute for it)
                         "frac_six_product_equiv": achieved * X3_MFMA_PER_PRODUCT / BF16_PEAK_TFLOPS,
                         # the same launches priced as fp32 work against the fp32 matrix peak (bounded by 2.67, not 1)
                         "frac_fp32_equiv": achieved / MFMA_F32_PEAK_TFLOPS,
                         # HBM side of the same launches: algorithmic operand + result bytes (storage types as launched) /
                         # their HIP-event time, against 8 TB/s — the bound that matters in the f16 mode, where one fp16
                         # MFMA per product tile leaves the GEMMs memory-bound
                         "hbm": {"achieved": hbm_gbs, "peak": 8000.0, "unit": "GB/s", "frac": hbm_gbs / 8000.0},
                         "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch (mean over the dominant family's launches of a step)",
                         "traffic_source": traffic_src, "algorithmic_bytes_per_launch": alg_bytes,
                         "launches_per_step": n_dom,
                         "families": {"gemm_p3_kernel<prec 2>": fam_p2, "gemm_x3_kernel": fam_x3},
                         "x3_ms_per_step": x3_ms / prof_steps, "x3_gflop_per_step": x3_flops / 1e9 / prof_steps,
                         # the rest of the GEMM family: gather-mode 3x3 convs / tiny shapes on v_mfma_f32_32x32x2_f32
                         "fp32_pipe_kernel": {"launches_per_step": len(f32) / prof_steps, "ms_per_step": f32_ms / prof_steps,
                                              "gflop_per_step": f32_flops / 1e9 / prof_steps,
                                              "achieved_tflops": f32_flops / (f32_ms * 1e-3) / 1e12 if f32_ms > 0 else 0.0,
                                              "peak": MFMA_F32_PEAK_TFLOPS},
                         "gemm_ms_per_step": gemm_ms / prof_steps,
                         "gemm_gflop_per_step": gemm_flops / 1e9 / prof_steps,
                         "gemm_family_tflops_fp32_equiv": achieved_all,
                         "measured": f"{prof_steps} EAGER steps after the timed region, HIP events around every launch on its "
                                     "stream (events cannot be read inside the replayed graph the headline number comes "
                                     "from; these steps launch every product by itself — the replayed step sends the data and "
                                     "weight gradient of a 1x1 conv out as ONE grid of the planes kernel, ud_gemm_p3_pair, "
                                     "which is ~0.5 ms per step faster than the sum timed here); profiles/r06/ holds the rocprofv3 --kernel-trace --stats summary of the "
                                     "graph-replayed steps of this same command (tools/gpu_round.sh), and "
                                     "tools/roofline_from_rocprof.py recomputes these numbers from it"},
        }
        # the same quantities from the rocprofv3 summary of the graph-REPLAYED steps of this command (committed under profiles/,
        # printed only when stamped with this tree's GEMM sources): what `frac` above under-states by timing eager, un-paired launches
        rep = _committed(args, bs, "roofline_replayed.json", "gemm_src_sha", _gemm_src_sha())
        line["roofline"]["replayed"] = rep
        # per kernel CLASS: launches, ms, GB moved and GB/s from the PMC passes (tools/hbm_bw_table.py --json)
        cls = _committed(args, bs, "hbm_classes.json", "csrc_sha", _csrc_sha())
        line["roofline"]["classes"] = cls
        if dp_diag is not None:
            line["data_parallel"] = dp_diag
        if args.dtype == "f16":
            # informational f16 line: the HBM side is the roofline (bound "hbm"); the matrix-pipe numbers move to "mfma"
            r = line["roofline"]
            r["mfma"] = {k: r[k] for k in ("achieved", "peak", "unit", "frac")}
            r.update(achieved=hbm_gbs, peak=8000.0, unit="GB/s", frac=hbm_gbs / 8000.0)
        if (args.model, args.size) == ("UDEB4", 256):
            # whole-step fractions from the algorithmic work per image (fwd+bwd, fp32):
            # 68.9 GFLOP and 1044 MB at bs 32 — per GPU, against the fp32 matrix peak and 8 TB/s
            ips = bs * args.steps / elapsed
            line["roofline"]["step_achieved_mfma_frac"] = 68.9e9 * ips / (MFMA_F32_PEAK_TFLOPS * 1e12)
            # B_alg per image: 996.3 MB of activations (fp32) + 3*4*P/bs of weights and weight gradients; half storage of the
            # trunk halves the activation term (~0.52 GB/img at bs 64)
            act_mb = 996.3 * (0.5 if (args.dtype == "f16" and args.storage == "f16") else 1.0)
            b_alg = (act_mb + 3 * 4 * 128.31 / bs) * 1e6
            line["roofline"]["step_algorithmic_bytes_per_image"] = b_alg
            line["roofline"]["step_achieved_hbm_frac"] = b_alg * ips / 8e12
        if world == 1 and not args.no_extra and (args.model, args.size, bs, args.dtype) == ("UDEB4", 256, 32, "f32"):
            torch.cuda.empty_cache()
            line["extra"] = extra_measurements()
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
    if dist.is_initialized():
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner through C stdio (buffered until exit): flush it first so that the JSON line is
        # the LAST line of stdout
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
