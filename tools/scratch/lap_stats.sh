#!/bin/bash
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/lapstats
rocprofv3 --kernel-trace --stats -d /tmp/lapstats -o t --output-format csv -- python $root/tools/train_bench.py arap_lap 12 > $root/gpurun_out/lap_stats.log 2>&1
cd $root
python - <<PY
import csv, glob
f = glob.glob("/tmp/lapstats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:26]:
    print(f"{r['Name'][:90]:90s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} ms avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
