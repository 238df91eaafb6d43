"""Per-replay kernel statistics from two kernel-trace summaries of the same script run with N1 and N2 replays:
(total(N2) - total(N1)) / (N2 - N1) per kernel.  usage: replay_stats.py stats_N1.csv N1 stats_N2.csv N2 OUT.csv"""
import csv, sys
a, n1, b, n2, out = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5]
def load(p):
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(p))}
A, B = load(a), load(b)
rows = []
for k, (c2, t2) in B.items():
    c1, t1 = A.get(k, (0, 0.0))
    dc, dt = (c2 - c1) / (n2 - n1), (t2 - t1) / (n2 - n1)
    if dc > 1e-9:
        rows.append((dt, k, dc))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
with open(out, "w") as f:
    f.write("kernel,launches_per_replayed_step,us_per_step,avg_us,percent\n")
    for dt, k, dc in rows:
        f.write(f'"{k[:160]}",{dc:.2f},{dt / 1e3:.2f},{dt / dc / 1e3:.2f},{100 * dt / tot:.2f}\n')
    f.write(f'"TOTAL kernel time per replayed step",{sum(r[2] for r in rows):.1f},{tot / 1e3:.1f},,100\n')
print(f"{out}: {tot / 1e3:.1f} us of kernels per step in {sum(r[2] for r in rows):.0f} launches")
