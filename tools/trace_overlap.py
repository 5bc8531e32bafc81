"""Timeline accounting of a rocprofv3 --kernel-trace csv of bench.py: for the LAST `steps` graph replays, how much of the step
some kernel is running (union), how much two run at once (sum - union), the idle time between kernels, per queue.
usage: python3 tools/trace_overlap.py <kernel_trace.csv> <steps> [ms_per_step]"""
import csv, sys, collections
path, steps = sys.argv[1], int(sys.argv[2])
rows = []
with open(path) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
rows.sort()
# the timed steps are the last `steps` replays: find them as the last steps * n kernels, n = launches per replay, from the
# periodicity of one marker kernel (the first kernel name of the tail)
names = [r[3] for r in rows]
tail = rows[-1][3]
cnt = collections.Counter(names)
# launches per step: the count of the most frequent GEMM kernel name in the last 1/3 of the trace divided by its per-step count is unknown,
# so instead: take the time window of the last `steps` * ms_per_step if given, else the last 40 % of the trace
if len(sys.argv) > 3:
    win = float(sys.argv[3]) * 1e6 * steps
else:
    win = 0.4 * (max(r[1] for r in rows) - rows[0][0])
# the timed replays are the densest stretch of the trace (eager warm-up / instrumented steps are host-paced): slide the window
best, j, busy = (-1, 0), 0, 0
for i in range(len(rows)):
    while j < len(rows) and rows[j][0] < rows[i][0] + win:
        busy += rows[j][1] - rows[j][0]
        j += 1
    if busy > best[0]:
        best = (busy, i, j)
    busy -= rows[i][1] - rows[i][0]
sel = rows[best[1]:best[2]]
t_end = max(r[1] for r in sel)
span = t_end - sel[0][0]
tot = sum(e - s for s, e, _, _ in sel)
# union and idle
ev = sorted((s, e) for s, e, _, _ in sel)
union, cur_s, cur_e = 0, ev[0][0], ev[0][1]
gaps = []
for s, e in ev[1:]:
    if s > cur_e:
        union += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
perq = collections.defaultdict(lambda: [0, 0])
for s, e, q, _ in sel:
    perq[q][0] += e - s
    perq[q][1] += 1
print("window %.2f ms (%d kernels, %.1f per step over %d steps)" % (span / 1e6, len(sel), len(sel) / steps, steps))
print("sum of kernel durations %.2f ms/step, some kernel running %.2f ms/step, two at once %.2f ms/step, idle %.2f ms/step in %.0f gaps/step (mean %.2f us)"
      % (tot / 1e6 / steps, union / 1e6 / steps, (tot - union) / 1e6 / steps, (span - union) / 1e6 / steps, len(gaps) / steps,
         (sum(gaps) / max(len(gaps), 1)) / 1e3))
gaps.sort()
if gaps:
    print("gap percentiles us: p50 %.2f p90 %.2f p99 %.2f max %.1f; gaps > 5 us: %d/step, %.2f ms/step" % (
        gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * .9)] / 1e3, gaps[int(len(gaps) * .99)] / 1e3, gaps[-1] / 1e3,
        sum(1 for g in gaps if g > 5000) / steps, sum(g for g in gaps if g > 5000) / 1e6 / steps))
for q, (t, n) in sorted(perq.items(), key=lambda kv: -kv[1][0]):
    print("queue %s: %.2f ms/step in %.0f kernels/step" % (q, t / 1e6 / steps, n / steps))

# launches of the replayed step by kernel family
import re
fam = collections.defaultdict(lambda: [0, 0])
for s_, e_, _, n in sel:
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    base = re.split(r"[<(]", n)[0]
    if base.startswith("at::native") or "rocclr" in base:
        base = "torch / rocclr: " + base
    fam[base][0] += e_ - s_
    fam[base][1] += 1
tor = sum(v[1] for k, v in fam.items() if k.startswith("torch / rocclr")) / steps
tort = sum(v[0] for k, v in fam.items() if k.startswith("torch / rocclr")) / 1e6 / steps
print("torch / rocclr kernels in the replayed step: %.0f launches, %.2f ms" % (tor, tort))
for k, (t, c) in sorted(fam.items(), key=lambda kv: -kv[1][0])[:40]:
    print("%8.3f ms %7.1f launches  %s" % (t / 1e6 / steps, c / steps, k[:90]))
