#!/bin/bash
# effective shader clock under the weight-gradient kernel: GRBM_GUI_ACTIVE per dispatch / kernel duration
out=${1:-gpurun_out/wgu_clock}; mkdir -p $out; root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $root/$out/pmc -o p --output-format csv -- python $root/tools/scratch/wgrad_probe.py pmc > $root/$out/pmc.log 2>&1
cd $root
python - <<PY
import csv, glob, collections
f = glob.glob("$out/pmc/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "wgrad_u_k" in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:40] + "/" + r.get("Grid_Size", "")].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for k, v in agg.items():
    v = v[3:]
    cyc = sum(a for a, _ in v) / len(v); ns = sum(b for _, b in v) / len(v)
    print(k, "cycles %.0f  ns %.0f  GHz %.3f" % (cyc, ns, cyc / ns))
PY
rm -rf $out/pmc
