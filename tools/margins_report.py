"""Fold the parity-bar records of several full GPU-suite runs (tests/margins.py -> gpurun_out/margins/run_<tag>_<pid>.jsonl)
into one table: per test the quantity that came closest to its bar — largest observed value over the runs, the bar, the
margin (bar / observed) — and whether the observed values were IDENTICAL in every run (the suite runs in deterministic mode).
usage: python tools/margins_report.py <out.md> <tag> [<tag> ...]"""
import glob
import json
import os
import sys
from collections import defaultdict


def load(tag):
    rows = []
    for f in glob.glob(os.path.join("gpurun_out", "margins", f"run_{tag}_*.jsonl")):
        with open(f) as fh:
            rows += [json.loads(ln) for ln in fh if ln.strip()]
    return rows


def main():
    out, tags = sys.argv[1], sys.argv[2:]
    per_run = {t: load(t) for t in tags}
    # (test, name) -> {tag: [observed ...]}
    obs, bars = defaultdict(lambda: defaultdict(list)), {}
    for t, rows in per_run.items():
        for r in rows:
            k = (r["test"], r["name"])
            obs[k][t].append(r["observed"])
            bars[k] = r["bar"]
    tests = defaultdict(list)
    for (test, name), by_tag in obs.items():
        worst = max(max(v) for v in by_tag.values())
        same = len({tuple(v) for v in by_tag.values()}) == 1 and len(by_tag) == len(tags)
        bar = bars[(test, name)]
        ratio = worst / bar if bar > 0 else (0.0 if worst == 0 else float("inf"))        # bar 0: an exact comparison
        tests[test].append((ratio, name, worst, bar, same))
    lines = [f"# Parity bars of the GPU suite: observed maxima over {len(tags)} full runs ({', '.join(tags)})", "",
             f"{sum(len(v) for v in tests.values())} recorded quantities in {len(tests)} tests.  Per test: the quantity closest "
             "to its bar.  `same`: every recorded quantity of the test had bit-identical values in all runs.", "",
             "| test | quantity closest to its bar | observed (max over runs) | bar | margin | same in all runs |",
             "|---|---|---|---|---|---|"]
    n_same = 0
    tight = []
    for test in sorted(tests):
        qs = sorted(tests[test], reverse=True)
        ratio, name, worst, bar, _ = qs[0]
        all_same = all(q[4] for q in qs)
        n_same += all_same
        margin = (1.0 / ratio) if ratio > 0 else float("inf")
        if margin < 4.0:
            tight.append((margin, test, name))
        lines.append(f"| `{test.replace('tests/', '')}` | {name} | {worst:.3g} | {bar:.3g} | "
                     f"{'inf' if margin == float('inf') else '%.1fx' % margin} | {'yes' if all_same else 'NO'} |")
    lines += ["", f"Tests whose every recorded quantity was identical in all runs: {n_same} / {len(tests)}.", ""]
    if tight:
        lines.append("Bars with less than 4x margin (each explained in profiles/README.md):")
        lines += [f"* {m:.1f}x  `{t.replace('tests/', '')}` — {n}" for m, t, n in sorted(tight)]
    with open(out, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print(f"{out}: {len(tests)} tests, {n_same} identical across runs, {len(tight)} with margin < 4x")


if __name__ == "__main__":
    main()
