import csv, sys
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        d[r["Name"]] = (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3)
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
n = float(sys.argv[3])
rows = []
for k in set(a) | set(b):
    ca, ta = a.get(k, (0, 0)); cb, tb = b.get(k, (0, 0))
    rows.append(((tb - ta) / n, k.replace("(anonymous namespace)::", "")[:90], ca / n, ta / max(ca, 1), cb / n, tb / max(cb, 1)))
rows.sort(key=lambda r: -abs(r[0]))
print("total us/step: a %.1f  b %.1f" % (sum(v[1] for v in a.values()) / n, sum(v[1] for v in b.values()) / n))
for d, k, ca, aa, cb, ab in rows[:int(sys.argv[4]) if len(sys.argv) > 4 else 20]:
    print(f"{d:8.1f} us/step  {k:90s} a {ca:6.1f} x {aa:7.2f}  b {cb:6.1f} x {ab:7.2f}")
