#!/bin/bash
# rocprofv3 kernel stats of the default bench into gpurun_out/$1 and a per-kernel ms/step table (18 executed steps)
tag=${1:-stats}
out=gpurun_out/$tag
mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench.log 2>&1
find $out/prof -name "*kernel_trace.csv" -delete
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
cp "$f" $out/kernel_stats.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$out/kernel_stats.csv")))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("total kernel ms/step %.2f" % (tot / 18e6))
for r in rows[:${2:-45}]:
    print("%6.3f ms/step %5.1f%% %6d calls %8.1f us  %s" % (int(r["TotalDurationNs"]) / 18e6, float(r["Percentage"]), int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY
grep -h '^{"metric"' $out/bench.log | cut -c1-160
