#!/bin/bash
# Full GPU suite on the box the way the driver runs it (python -m pytest tests/ -x -q -m gpu), with every parity bar
# recorded (tests/margins.py).  usage: tools/gpu_suite.sh <tag> [all]   ("all": no -x, every failure is listed)
tag=${1:-s1}
out=gpurun_out/$tag
mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1 UD_MARGIN_RUN=$tag UD_MARGIN_DIR=$PWD/gpurun_out/margins
(rocminfo | grep -E "Marketing" | head -2; hostname; date) > $out/env.log 2>&1
xflag="-x"; [ "$2" == "all" ] && xflag=""
timeout 2400 python -m pytest tests/ $xflag -q -m gpu -rA --durations=15 --timeout 1200 > $out/pytest_gpu.log 2>&1
echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed" $out/pytest_gpu.log | tail -15
