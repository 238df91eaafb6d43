# which slots of a plan's table keep changing in the swap path (faust_swap: a cycle of four pairs)?
import os, sys, struct, collections, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import train_bench
from surfacenetworks_amd import plans
log = collections.defaultdict(list)
real = plans.Plan._launch
def spy(self, b):
    log[id(self)].append(tuple(b))
    return real(self, b)
plans.Plan._launch = spy
plans.set_graphs(False)
train_bench.faust_swap("cuda", steps=40)
for pid, rows in list(log.items())[:6]:
    n = len(rows[0]); tail = rows[-64:]
    distinct = [len({r[j] for r in tail}) for j in range(n)]
    print("plan", pid % 10000, "runs", len(rows), "distinct values per slot over the last 64 runs:", distinct, "distinct tables:", len(set(tail)))
