"""One replayed step of a rocprofv3 --kernel-trace csv of bench.py as an ordered launch list: index, start offset (us), duration (us),
grid, workgroup, kernel name — the step runs on one queue, so the order is the tape's.  The step is cut at the one
`split_h2_multi_kernel` launch every forward starts with (the batch split of all weight planes); the LAST complete step of the
trace's densest stretch (the timed hipGraph replays) is written, with the median duration of each position over the last
`steps` replays.  usage: python3 tools/step_sequence.py <kernel_trace.csv> <out.txt> [steps=8] [marker kernel=split_h2_multi_kernel]"""
import csv
import re
import statistics
import sys

path, out = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
marker = sys.argv[4] if len(sys.argv) > 4 else "split_h2_multi_kernel"      # a kernel launched exactly once per step
rows = []
with open(path) as fh:
    for r in csv.DictReader(fh):
        g = [int(r.get(k, 0) or 0) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")]
        w = [int(r.get(k, 0) or 0) for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z")]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], g, w))
rows.sort(key=lambda t: t[0])
marks = [i for i, r in enumerate(rows) if marker in r[2]]
segs = [(a, b) for a, b in zip(marks[:-1], marks[1:])]
# replayed steps are the shortest segments in wall time with the modal launch count
from collections import Counter
modal = Counter(b - a for a, b in segs).most_common(1)[0][0]
segs = [s for s in segs if s[1] - s[0] == modal]
segs.sort(key=lambda s: rows[s[1]][0] - rows[s[0]][0])
use = segs[:steps]
n = modal


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(?:<[^(]*>)?)", name)
    name = m.group(1) if m else name
    name = re.sub(r"at::native::", "aten::", name)
    return name[:110]


with open(out, "w") as fh:
    a0 = use[0][0]
    t0 = rows[a0][0]
    tot = 0.0
    fh.write("# %d launches per step; durations: median over %d replayed steps; step wall %.3f ms\n" %
             (n, len(use), (rows[use[0][1]][0] - t0) / 1e6))
    fh.write("# idx start_us dur_us grid wg name\n")
    for k in range(n):
        durs = [(rows[a + k][1] - rows[a + k][0]) / 1e3 for a, _ in use]
        r = rows[a0 + k]
        d = statistics.median(durs)
        tot += d
        fh.write("%4d %9.1f %7.2f %8d %4d %s\n" % (k, (r[0] - t0) / 1e3, d, r[3][0] * max(r[3][1], 1) * max(r[3][2], 1) // max(r[4][0] * max(r[4][1], 1) * max(r[4][2], 1), 1),
                                                   r[4][0] * max(r[4][1], 1) * max(r[4][2], 1), short(r[2])))
    fh.write("# sum of kernel durations %.3f ms\n" % (tot / 1e3))
