"""Condense rocprofv3 --pmc counter_collection.csv files into a small per-kernel summary CSV for profiles/.
usage: python tools/summarize_pmc.py OUT.csv LAUNCHES_PER_PHASE IN1.csv [IN2.csv ...]"""
import collections
import csv
import re
import sys


def short(name):
    m = re.search(r"(spmm_\w+<[^>]*>|\w+_k<[^>]*>|\w+)\(", name.replace("void (anonymous namespace)::", ""))
    return m.group(1) if m else name[:60]


def main():
    out, per_phase, files = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
    agg = collections.OrderedDict()
    for fn in files:
        for r in csv.DictReader(open(fn)):
            if "spmm" not in r["Kernel_Name"]:
                continue
            agg.setdefault((short(r["Kernel_Name"]), r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    with open(out, "w") as f:
        f.write("kernel,counter,phase,launches,mean_per_launch\n")
        for (k, c), v in agg.items():
            for i in range(0, len(v), per_phase):
                ch = v[i: i + per_phase]
                f.write(f'"{k}",{c},{i // per_phase},{len(ch)},{sum(ch) / len(ch):.1f}\n')


if __name__ == "__main__":
    main()
