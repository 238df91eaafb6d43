#!/bin/bash
# kernel statistics of a tools/train_bench.py configuration: cfg_stats.sh CONFIG STEPS
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cfgstats
rocprofv3 --kernel-trace --stats -d /tmp/cfgstats -o t --output-format csv -- python $root/tools/train_bench.py $1 ${2:-12} > $root/gpurun_out/cfg_stats.log 2>&1
cd $root
python - <<PY
import csv, glob
f = glob.glob("/tmp/cfgstats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot / 1e6, "kernels", sum(int(r['Calls']) for r in rows))
for r in rows[:60]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} ms avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
