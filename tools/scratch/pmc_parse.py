import csv, glob, collections, re
for c in ("TCC_EA0_RDREQ_sum","TCC_EA0_WRREQ_sum"):
    f = glob.glob(f"gpurun_out/pmc_{c}/*/*counter_collection.csv")
    rows = list(csv.DictReader(open(f[0])))
    agg = collections.defaultdict(list)
    for r in rows:
        m = re.search(r"(spmm_q3_lds\w*|spmm_stats_reduce_k)", r["Kernel_Name"])
        if m:
            agg[(m.group(1), r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
    mult = 128 if "RD" in c else 64
    for k, v in sorted(agg.items()):
        print(c, k, len(v), round(sum(v)/len(v)*mult/1e6, 1), "MB")
