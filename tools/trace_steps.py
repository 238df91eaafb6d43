"""Per-step GPU busy / idle breakdown from a rocprofv3 --kernel-trace CSV of bench.py (steps are delimited by the
first blockdiag_rowptr launch of each sample_batch)."""
import collections
import csv
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n).replace("at::native::", "")
    return n[:72]


def main(path, dump=None):
    rows = list(csv.DictReader(open(path)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
    idx = [i for i, e in enumerate(ev) if "blockdiag_rowptr" in e[2]]
    starts = [idx[0]]
    for a, b in zip(idx, idx[1:]):
        if ev[b][0] - ev[a][0] > 5e6:
            starts.append(b)
    print("steps in trace:", len(starts) - 1)
    for s0, s1 in list(zip(starts, starts[1:]))[-5:]:
        sub = ev[s0:s1]
        wall = (ev[s1][0] - ev[s0][0]) / 1e6
        busy = sum(e[1] - e[0] for e in sub) / 1e6
        print(f"step wall {wall:.2f} ms  busy {busy:.2f} ms  kernels {len(sub)}")
    s0, s1 = starts[-3], starts[-2]
    sub = ev[s0:s1]
    agg = collections.defaultdict(lambda: [0, 0])
    for e in sub:
        agg[short(e[2])][0] += e[1] - e[0]
        agg[short(e[2])][1] += 1
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:32]:
        print(f"{k:72s} {v[1]:4d} {v[0] / 1e6:7.3f} ms")
    if dump:
        t0 = sub[0][0]
        with open(dump, "w") as fh:
            for e in sub:
                fh.write(f"{(e[0] - t0) / 1e3:9.1f} {(e[1] - e[0]) / 1e3:8.1f} {short(e[2])}\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
