"""GPU: where the HOST spends the two-pass train step — inside CUDAGraph.replay() (does hipGraphLaunch return before the
graph has run?) or in the Python between the replays (optimizer bookkeeping, perturbation) — against the step's wall time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import runpy

acc = {"replay": 0.0, "n": 0}
_orig = torch.cuda.CUDAGraph.replay


def timed(self):
    t0 = time.perf_counter()
    _orig(self)
    acc["replay"] += time.perf_counter() - t0
    acc["n"] += 1


torch.cuda.CUDAGraph.replay = timed
sys.argv = ["tools/bench_train_step.py", "32", "20"]
t0 = time.perf_counter()
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_train_step.py"), run_name="__main__")
print("replay() calls %d, host time inside them %.1f ms per call" % (acc["n"], acc["replay"] / max(acc["n"], 1) * 1e3))
