"""Every numeric parity bar of the GPU suite goes through `within()`: it records (test, quantity, observed, bar) as one
JSON line under gpurun_out/margins/ (UD_MARGIN_DIR) and returns observed <= bar.  tools/margins_report.py folds the files of
several full runs into profiles/rNN/margins.md — per test the largest observed value, its bar and the margin — which is how
the suite shows that its bars sit well above what the kernels actually do (and, run to run, that the observed values do not
move: the default mode is deterministic)."""
import json
import math
import os

_DIR = os.environ.get("UD_MARGIN_DIR") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                        "gpurun_out", "margins")
_FH = None


def _fh():
    global _FH
    if _FH is None:
        os.makedirs(_DIR, exist_ok=True)
        _FH = open(os.path.join(_DIR, "run_%s_%d.jsonl" % (os.environ.get("UD_MARGIN_RUN", "x"), os.getpid())), "a")
    return _FH


def record(name, observed, bar):
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    try:
        fh = _fh()
        fh.write(json.dumps({"test": test, "name": name, "observed": float(observed), "bar": float(bar)}) + "\n")
        fh.flush()
    except OSError:
        pass


def within(name, observed, bar):
    """Record and compare.  NaN never passes."""
    observed = float(observed)
    record(name, observed, bar)
    return (not math.isnan(observed)) and observed <= bar
