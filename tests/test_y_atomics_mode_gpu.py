"""GPU: the package DEFAULT setting of cfg.deterministic.  The suite runs with split-K GEMMs adding their partial products in
a fixed order (tests/conftest.py pins cfg.deterministic = True so that its results are a function of the code alone); the
default (False) lets them accumulate with fp32 atomics instead — one launch less per split GEMM, 2.3 ms of the bs-32 step,
results that differ in the last bits from run to run.  Here that mode
is held to the default mode's result on the UDEB4 train step (N = 4, the 504 parameter gradients) with bars that follow the
conditioning of what is compared (tests/test_z_fused_selfcheck_gpu.py: tensors against their largest entry, the scalar gate
gradients against the sum of their terms' magnitudes), and two atomics runs to each other.  Kernel level: both forms of every
(tile, split-K) plan against float64 in tests/test_a_kernels_gpu.py::test_gemm_every_tile_configuration_and_split.
Reference: the step of engine/abstract_engine.py:207-281 on model/unidefense.py:174-256."""
import pytest
import torch

from tests.margins import within
from tests.test_z_fused_selfcheck_gpu import _dev, _grad_bars, _rel, _run

pytestmark = pytest.mark.gpu


def test_atomics_mode_equals_default_mode_within_conditioning():
    from unidefense_amd.config import override
    dev = _dev()
    with override(deterministic=True):                                          # (the suite's setting, made explicit)
        _, _, _, _, _, _, cond = _run(dev, False, 0.0, 4, 11, False)           # operator path: the gates' conditioning sums
        l_det, o_det, f_det, g_det, _, _, _ = _run(dev, True, 0.0, 4, 11, False)
    with override(deterministic=False):
        l_a1, o_a1, f_a1, g_a1, _, _, _ = _run(dev, True, 0.0, 4, 11, False)
        l_a2, o_a2, f_a2, g_a2, _, _, _ = _run(dev, True, 0.0, 4, 11, False)
    assert within("loss: atomics vs ordered", _rel(l_a1, l_det), 1e-5)       # observed 4e-7 .. 1.1e-6, run to run
    for k in ("cls_out", "rec"):
        assert within(f"{k}: atomics vs ordered", _rel(o_a1[k], o_det[k]), 1e-4)      # 1/10 of the 1e-3 parity bar
    worst_f = max(_rel(f_a1[k], f_det[k]) for k in f_det)
    assert within("stage outputs: atomics vs ordered", worst_f, 1e-4)
    bad = _grad_bars(g_det, g_a1, cond, "atomics vs ordered")
    bad += _grad_bars(g_a1, g_a2, cond, "atomics run 1 vs run 2")
    assert not bad, bad[:10]
