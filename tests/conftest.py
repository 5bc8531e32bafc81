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


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
