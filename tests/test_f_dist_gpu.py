"""GPU, one device: the data-parallel code path (RCCL process group, SyncBatchNorm statistics exchange, bucketed
gradient all-reduce, their capture into the step's hipGraph) exercised in a world of ONE process with
UD_FORCE_COLLECTIVES=1.  With a single rank every collective is the identity, so the step must reproduce the plain
single-GPU step: same loss after the same number of steps.  (The N>1 semantics are covered on CPU with gloo in
tests/test_parallel_cpu.py; N>1 on GPUs is only ever run by the driver's scaling bench.)"""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(extra_env, *flags):
    env = dict(os.environ, **extra_env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "8",
           "--no-cpu-baseline", *flags]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    # stdout is the ONE JSON line, and it comes last (RCCL's version banner is flushed before it)
    assert lines and lines[-1].startswith("{") and sum(ln.startswith("{") for ln in lines) == 1, lines[-3:]
    return json.loads(lines[-1]), r.stderr


def test_bench_line_contract():
    """The ONE JSON line bench.py prints: keys and types the driver reads, the roofline object, exec mode."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    line, err = _bench({}, )
    for k, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                   ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str),
                   ("data", str), ("config", dict), ("roofline", dict)):
        assert isinstance(line[k], typ), (k, line.get(k))
    assert line["vs_baseline"] is None and line["scaling"] == "weak" and line["dtype"] == "f32"
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1
    assert "workload" in line["config"] and line["config"]["exec"] == "hipgraph", err[-1500:]
    assert abs(line["value"] - 8 * 1e3 / line["ms_per_step"]) < 1e-6 * line["value"]          # bs 8 in _bench
    r = line["roofline"]
    # the matrix-pipe GEMM family is priced against ITS pipe: peak = 2500 TFLOP/s dense 16-bit / the MFMAs its launches execute
    # per algorithmic fp32 product (3 on pre-split fp16 planes, 6 with the in-kernel bf16 split; the mix is the step's)
    mpp = r["mfma_per_product"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and 3.0 <= mpp <= 6.0 and abs(r["peak"] - 2500.0 / mpp) < 1e-9
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] < 1.0
    assert abs(r["executed_mfma_tflops"] - mpp * r["achieved"]) < 1e-6 and r["pipe_peak"] == 2500.0
    assert abs(r["frac_of_pipe"] - r["frac"]) < 1e-12 and abs(r["frac_fp32_equiv"] - r["achieved"] / 157.3) < 1e-9
    assert abs(r["frac_six_product_equiv"] - 6 * r["achieved"] / 2500.0) < 1e-9
    fams = r["families"]
    assert set(fams) == {"gemm_p3_kernel<prec 2>", "gemm_x3_kernel"}
    assert fams["gemm_p3_kernel<prec 2>"]["mfma_per_product"] == 3 and fams["gemm_x3_kernel"]["mfma_per_product"] == 6
    assert "traffic" in r and line["config"]["n_ranks_seen"] == 1
    assert "data_parallel" not in line                                   # a plain one-rank run issues no collective
    # the same line with the data-parallel path forced on (RCCL process group of one rank): the diagnostics that make the
    # first real N > 1 run readable — how many SyncBN sums, their device time, the exposed tail of the gradient exchange
    forced, err = _bench({"UD_FORCE_COLLECTIVES": "1", "MASTER_PORT": "29543"})
    d = forced["data_parallel"]
    for k in ("syncbn_exchanges_per_step", "syncbn_exchange_ms", "allreduce_exposed_ms", "allreduce_bytes", "allreduce_collectives"):
        assert isinstance(d[k], float) and d[k] >= 0.0, (k, d)
    assert 150 <= d["syncbn_exchanges_per_step"] <= 260, d              # 99 BatchNorms: forward + backward sums
    assert 0.0 < d["syncbn_exchange_ms"] < 10.0, d
    assert abs(d["allreduce_bytes"] - 4 * 128.31e6) < 2e6, d             # every trainable parameter's gradient, fp32, once
    assert 20 <= d["allreduce_collectives"] <= 80, d                     # ~34 in-place spectral weights + the packed buckets


@pytest.mark.parametrize("flags", [(), ("--eager",)])
def test_forced_collectives_match_plain_step(flags):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    plain, _ = _bench({"UD_FORCE_COLLECTIVES": "0"}, *flags)
    forced, err = _bench({"UD_FORCE_COLLECTIVES": "1", "MASTER_PORT": "29541"}, *flags)
    print("  plain :", plain["config"]["exec"], plain["config"]["final_loss"], "%.1f ms" % plain["ms_per_step"])
    print("  forced:", forced["config"]["exec"], forced["config"]["final_loss"], "%.1f ms" % forced["ms_per_step"])
    a, b = plain["config"]["final_loss"], forced["config"]["final_loss"]
    from tests.margins import within
    assert within("forced collectives vs plain: final loss", abs(a - b) / abs(a), 1e-4), (a, b)
    # gradients: streamed buckets (views of the reduced flat buffers) must be the plain step's gradients
    ga, gb = plain["config"]["grad_l1"], forced["config"]["grad_l1"]
    print("  grad L1:", ga, gb)
    assert ga > 0 and within("forced collectives vs plain: gradient L1", abs(ga - gb) / ga, 1e-4), (ga, gb)
    if not flags:      # the RCCL calls must survive hipGraph capture, else the 8-GPU bench would run eagerly
        assert forced["config"]["exec"] == "hipgraph", err[-2000:]
