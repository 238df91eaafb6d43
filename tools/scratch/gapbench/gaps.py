"""end -> start gaps between consecutive kernels of a rocprofv3 kernel trace, grouped by (previous kernel, next kernel, grids)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0)) for r in rows)
agg = collections.OrderedDict()
for a, b in zip(ev, ev[1:]):
    gap = (b[0] - a[1]) / 1e3
    if gap > 100:        # a stream synchronisation between sequences
        continue
    agg.setdefault((a[2], a[3], b[2], b[3]), []).append((gap, (a[1] - a[0]) / 1e3, (b[1] - b[0]) / 1e3))
for k, v in agg.items():
    g = sorted(x[0] for x in v)
    print(f"{k[0]:12s} grid {k[1] // 256:6d} wg -> {k[2]:12s} grid {k[3] // 256:6d} wg: n {len(v):3d} gap median {g[len(g) // 2]:6.2f} us (min {g[0]:5.2f} max {g[-1]:5.2f}); durations {sum(x[1] for x in v) / len(v):7.1f} -> {sum(x[2] for x in v) / len(v):7.1f} us")
