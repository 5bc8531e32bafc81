#!/bin/bash
# The full GPU suite with the package DEFAULT (fp32 atomics in the split-K GEMMs) instead of the deterministic mode the suite pins:
# shows that the bars hold under run-to-run noise too.  Not -x: every failure is listed.   usage: tools/gpu_suite_atomics.sh <tag>
tag=${1:-a1}
out=gpurun_out/$tag
mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1 UD_TEST_DETERMINISTIC=0 UD_MARGIN_RUN=$tag UD_MARGIN_DIR=$PWD/gpurun_out/margins
timeout 2400 python -m pytest tests/ -q -m gpu -rA --timeout 1200 > $out/pytest_gpu.log 2>&1
echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed" $out/pytest_gpu.log | tail -15
