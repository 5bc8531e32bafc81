"""GPU-box host probe: cores visible / allowed, and oracle step time vs torch thread count (bounded)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    th, bs = int(sys.argv[1]), int(sys.argv[2])
    sys.path.insert(0, ROOT)
    import torch
    torch.set_num_threads(th)
    from oracle import param_fill
    from tests import oracle_util as ou
    x = param_fill.make_input(bs, 256, 0); tgt = param_fill.make_labels(bs); rng = ou.make_rng(bs, 1, 0.5)
    sd = ou.oracle_state(-10.0, 0.0, requires_grad=True)
    t0 = time.perf_counter(); ou.oracle_train_pass1(sd, x, tgt, rng, 0.5); t1 = time.perf_counter()
    for v in sd.values(): v.grad = None
    ou.oracle_train_pass1(sd, x, tgt, rng, 0.5); t2 = time.perf_counter()
    print(f"threads {th} bs {bs}: first {t1-t0:.2f}s second {t2-t1:.2f}s -> {bs/(t2-t1):.2f} img/s", flush=True)
    sys.exit(0)
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
print("OMP_NUM_THREADS", os.environ.get("OMP_NUM_THREADS"))
for th in (8, 16, 32, 64, 128):
    try:
        r = subprocess.run([sys.executable, __file__, str(th), "4"], capture_output=True, text=True, timeout=150)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
    except subprocess.TimeoutExpired:
        print(f"threads {th}: timeout 150 s")
