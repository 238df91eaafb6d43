"""Aggregate, for every blit kernel (copyBuffer / fillBuffer) in one training step of a rocprofv3 kernel trace, the kernels
launched just before and after.  One step = from one masked_sl1_fwd_k launch to the next."""
import collections
import csv
import glob
import sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0][:44]
names = [short(r["Kernel_Name"]) for r in rows]
marks = [i for i, n in enumerate(names) if n.startswith("masked_sl1_fwd_k")]
lo, hi = marks[-3], marks[-2]
agg = collections.Counter()
dur = collections.Counter()
blit = 0
for i in range(lo, hi):
    n = names[i]
    if "copyBuffer" in n or "fillBuffer" in n or "at::native" in rows[i]["Kernel_Name"]:
        full = rows[i]["Kernel_Name"]
        tag = next((t for t in ("FillFunctor", "direct_copy", "CatArray", "multi_tensor", "BinaryFunctor", "reduce_kernel", "index")
                    if t in full), "")
        key = (n[:22] + " " + tag, names[i - 1], names[i + 1])
        agg[key] += 1
        d = int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])
        dur[key] += d
        blit += d
span = int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[lo:hi])
print(f"step: {hi - lo} launches, span {span / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, blits {blit / 1e3:.1f} us")
for key, n in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(f"{n:4d} {dur[key] / n / 1e3:6.2f} us {key[0]:30s} after {key[1]:44s} before {key[2]}")
# gaps: idle time between consecutive launches
gaps = sorted(((int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"])) / 1e3, names[i], names[i + 1]) for i in range(lo, hi))
print("largest idle gaps (us):")
for g in gaps[-12:]:
    print(f"  {g[0]:8.2f}  {g[1]} -> {g[2]}")
print("sum of positive gaps: %.1f us" % sum(g[0] for g in gaps if g[0] > 0))
