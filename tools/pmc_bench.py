"""In-step HBM traffic of every kernel of the timed bench step, from rocprofv3 --pmc passes over `bench.py` ITSELF (not a
back-to-back probe): usage
    python tools/pmc_bench.py OUT_PREFIX RD_counter_collection.csv WR_counter_collection.csv [SKIP_DISPATCHES_FRACTION]
Writes OUT_PREFIX.json  {kernel: {launches, read_bytes_mean, write_bytes_mean, by_grid: {grid: {...}}}, "_step_total": ...}
   and OUT_PREFIX_raw.csv (one row per dispatch of the SpMM kernels: kernel, grid, RDREQ, WRREQ) as the raw evidence.
bytes = TCC_EA0_RDREQ_sum * 128 + TCC_EA0_WRREQ_sum * 64 — the gfx950 correction of MI355X_MICROARCH.md (every read request
of these kernels is a 128-byte line: TCC_EA0_RDREQ_32B_sum = 0; FETCH_SIZE tallies them at 64 bytes)."""
import collections
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def short(name):
    name = name.replace("void (anonymous namespace)::", "")
    m = re.match(r"([\w:]+(?:<[^(]*>)?)\(", name)
    return (m.group(1) if m else name)[:90]


def load(path, counter):
    rows = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            rows.append((int(r["Dispatch_Id"]), short(r["Kernel_Name"]), int(r["Grid_Size"]), float(r["Counter_Value"])))
    rows.sort()
    return rows


def main():
    out, rd_csv, wr_csv = sys.argv[1:4]
    # 4th argument: `last=N` keeps exactly the last N training steps; a number drops that fraction of all dispatches
    # (default: the first half — set-up and warm-up)
    arg = sys.argv[4] if len(sys.argv) > 4 else "0.5"
    if arg.startswith("last="):
        res, agg, rd, wr, n0, tot_r, tot_w = summarize(rd_csv, wr_csv, last_steps=int(arg[5:]))
        res["_steps_counted"] = int(arg[5:])
    else:
        res, agg, rd, wr, n0, tot_r, tot_w = summarize(rd_csv, wr_csv, float(arg))
    json.dump(res, open(out + ".json", "w"), indent=1)
    with open(out + "_raw.csv", "w") as fh:
        fh.write("dispatch,kernel,grid,TCC_EA0_RDREQ_sum,TCC_EA0_WRREQ_sum\n")
        wmap = {(d, k, g): v for d, k, g, v in wr}
        for d, k, g, v in rd[n0:]:
            if "spmm" in k:
                fh.write(f'{d},"{k}",{g},{v:.0f},{wmap.get((d, k, g), float("nan")):.0f}\n')
    top = sorted(((e["read_bytes"] + e["write_bytes"], k) for k, e in agg.items()), reverse=True)[:12]
    for b, k in top:
        print(f"{b / 1e9:9.3f} GB  {k}")
    print(f"total counted: read {tot_r / 1e9:.2f} GB write {tot_w / 1e9:.2f} GB")


def step_start(rows, last_steps):
    """Index of the first dispatch of the `last_steps`-th step from the end: a step starts with the batch assembly of
    sample_batch (a burst of blockdiag_rowptr launches); bursts more than 50 dispatches apart are different steps."""
    marks = [i for i, r in enumerate(rows) if "blockdiag_rowptr" in r[1]]
    starts = [m for j, m in enumerate(marks) if j == 0 or m - marks[j - 1] > 50]
    if len(starts) < last_steps:
        raise ValueError(f"only {len(starts)} steps found in the counter file, {last_steps} wanted")
    return starts[-last_steps]


def summarize(rd_csv, wr_csv, skip=0.5, last_steps=None):
    """The per-kernel table of two counter passes (read requests, write requests) of the same deterministic command.
    last_steps: keep exactly the dispatches of the last N training steps (else: drop the first `skip` of all dispatches)."""
    rd, wr = load(rd_csv, "TCC_EA0_RDREQ_sum"), load(wr_csv, "TCC_EA0_WRREQ_sum")
    if last_steps is not None:
        n0, w0 = step_start(rd, last_steps), step_start(wr, last_steps)
    else:
        n0, w0 = int(len(rd) * skip), int(len(wr) * skip)
    if [r[1:3] for r in rd] != [w[1:3] for w in wr]:
        # the two passes are two runs of the same deterministic command: align by (kernel, grid) order per kernel
        print("warning: dispatch sequences differ between the passes; aligning per kernel", file=sys.stderr)
    agg = collections.OrderedDict()
    per_k_wr = collections.defaultdict(list)
    for d, k, g, v in wr[w0:]:
        per_k_wr[(k, g)].append(v)
    per_k_rd = collections.defaultdict(list)
    for d, k, g, v in rd[n0:]:
        per_k_rd[(k, g)].append(v)
    tot_r = tot_w = 0.0
    for (k, g), rv in per_k_rd.items():
        wv = per_k_wr.get((k, g), [0.0])
        e = agg.setdefault(k, {"launches": 0, "read_bytes": 0.0, "write_bytes": 0.0, "by_grid": {}})
        rb, wb = sum(rv) * 128, sum(wv) * 64 * len(rv) / max(len(wv), 1)
        e["launches"] += len(rv)
        e["read_bytes"] += rb
        e["write_bytes"] += wb
        e["by_grid"][str(g)] = {"launches": len(rv), "read_bytes_mean": rb / len(rv), "write_bytes_mean": wb / len(rv)}
        tot_r += rb
        tot_w += wb
    res = {"_comment": "HBM traffic per launch, measured IN the training step: rocprofv3 --pmc TCC_EA0_RDREQ_sum / TCC_EA0_WRREQ_sum "
                       "(one counter per pass) over `python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary`; "
                       "bytes = RDREQ*128 + WRREQ*64 (gfx950 correction); the first half of the dispatches (set-up and warm-up) dropped",
           "_total_bytes_counted": {"read": tot_r, "write": tot_w}}
    from bench import csrc_digest           # the kernel sources these counters belong to (bench.py withholds them on a mismatch)

    res["_csrc_sha256"] = csrc_digest()
    for k, e in agg.items():
        res[k] = {"launches": e["launches"], "read_bytes_mean": e["read_bytes"] / e["launches"],
                  "write_bytes_mean": e["write_bytes"] / e["launches"], "by_grid": e["by_grid"]}
    return res, agg, rd, wr, n0, tot_r, tot_w


if __name__ == "__main__":
    main()
