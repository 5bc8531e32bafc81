"""A/B helper: set module-level knobs of the package, then run a script in this process.
usage: python tools/run_with.py kernels._XCD_CONTIGUOUS=True tape._DW_FUSED_ADD=False -- bench.py --steps 10
(the compile-time constants that used to be UD_* environment variables are plain module attributes now)"""
import ast
import importlib
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
args = sys.argv[1:]
cut = args.index("--")
for a in args[:cut]:
    target, value = a.split("=", 1)
    mod, attr = target.rsplit(".", 1)
    m = importlib.import_module("unidefense_amd." + mod)
    assert hasattr(m, attr), target
    setattr(m, attr, ast.literal_eval(value))
sys.argv = args[cut + 1:]
runpy.run_path(sys.argv[0], run_name="__main__")
