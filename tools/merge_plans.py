"""Merge measured GEMM plans (a UD_GEMM_TUNE_CACHE file written on an MI355X) into the shipped defaults
(unidefense_amd/gemm_plans_gfx950.json).  usage: python tools/merge_plans.py <cache.json> [prefix ...]
With prefixes, only keys whose kind starts with one of them are taken (e.g. `p2c`: the spectral-conv plane plans)."""
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "unidefense_amd", "gemm_plans_gfx950.json")
src, prefixes = sys.argv[1], tuple(sys.argv[2:])
with open(dst) as fh:
    base = json.load(fh)
with open(src) as fh:
    new = json.load(fh)
added = changed = 0
for k, v in new.items():
    kind = json.loads(k)[0]
    if prefixes and not str(kind).startswith(prefixes):
        continue
    if k not in base:
        added += 1
    elif base[k] != v:
        changed += 1
    base[k] = v
with open(dst, "w") as fh:
    json.dump(base, fh)
print(f"{dst}: {added} added, {changed} changed, {len(base)} entries")
