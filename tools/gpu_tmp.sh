#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | cut -c1-400
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-300
