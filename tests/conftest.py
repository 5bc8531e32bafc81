import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The suite's results must be a function of the code alone, not of the box or the run (unidefense_amd/config.py reads these
# at import; the child processes of the multi-process tests inherit them):
#  * split-K GEMMs add their partial products in a fixed order (cfg.deterministic; the package default is fp32 atomics);
#  * no on-line GEMM tuning: a shape outside the shipped plans takes the cost-model plan instead of whatever measured
#    fastest on this particular box (a different plan is a different summation order).  The tuner has its own test.
# tests/test_y_atomics_mode_gpu.py covers the other setting of the first switch.
os.environ["UD_DETERMINISTIC"] = os.environ.get("UD_TEST_DETERMINISTIC", "1")
os.environ.setdefault("UD_GEMM_TUNE", "0")

from tests import oracle_util as _ou      # noqa: E402

_ou.fit_cpu_threads()                      # the CPU oracle on the job's CPU quota, not on every logical CPU of the host


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


# ---- the two settings the product runs in --------------------------------------------------------------------------------
# "suite": what the rest of the suite pins (split-K partial products added in a fixed order, results a function of the code
#          alone);
# "bench": what `python bench.py` times — fp32 atomics for split-K, stream-K plans allowed — with the planes path of the 1x1
#          convs (kernels.spectral_*) forced on for every shape the kernel takes, so that the reference-golden / oracle tests
#          below also hold the configuration behind the headline number (and the fp16 x 2 planes GEMM at every stage) to the
#          reference, not only the deterministic one.
RUN_MODES = ("suite", "bench")


@pytest.fixture(params=RUN_MODES)
def run_mode(request):
    from unidefense_amd.config import override
    kw = dict(deterministic=True) if request.param == "suite" else dict(deterministic=False, spectral_p2="on")
    with override(**kw):
        yield request.param
