#!/bin/bash
# the three tests ABOUT the deterministic switch, under both suite settings
export PYTHONDONTWRITEBYTECODE=1 UD_MARGIN_RUN=modes
for d in 1 0; do
  echo "UD_TEST_DETERMINISTIC=$d: $(UD_TEST_DETERMINISTIC=$d python -m pytest tests/test_y_atomics_mode_gpu.py tests/test_y_fullsize_gpu.py -q -m gpu -k 'atomics or deterministic' 2>&1 | tail -1)"
done
