"""GPU box (CPU work): how long the float64 CPU oracle of the element-wise gradient tests takes with torch's default thread
count against the job's CPU quota (bench.host_cores) — the boxes show 256 logical CPUs under a 16-CPU quota."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from oracle import param_fill
from tests import oracle_util as ou

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "host_cores", bench.host_cores(),
      "torch default threads", torch.get_num_threads(), "OMP_NUM_THREADS", os.environ.get("OMP_NUM_THREADS"), flush=True)
try:
    print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip(), flush=True)
except OSError as e:
    print("cpu.max unreadable", e)


def run(n, dtype):
    x = param_fill.make_input(n, 256, 38)
    tgt = param_fill.make_labels(n)
    rng = ou.make_rng(n, 138, 0.5)
    sd = ou.oracle_state(0.0, 0.3, dtype=dtype, requires_grad=True)
    t0 = time.perf_counter()
    ou.oracle_train_pass1(sd, x.to(dtype), tgt, rng, 0.5, ou.LAMBDAS)
    return time.perf_counter() - t0


order = [int(a) for a in sys.argv[1:]] or [0, bench.host_cores(), 2 * bench.host_cores()]
default = torch.get_num_threads()
for th in order:
    torch.set_num_threads(th if th > 0 else default)
    print("threads %3d: float64 N=2 %.1f s, float32 N=2 %.1f s" % (torch.get_num_threads(), run(2, torch.float64), run(2, torch.float32)),
          flush=True)
