"""Vertex / face ORDER probe: the config-5 products and the config-3 step on grid-order, permuted and re-ordered meshes
(bench.py's `secondary.products[*].order`, `secondary.config3_order`) without the rest of the bench.
Usage: python tools/order_probe.py [c5] [c3] [order=grid|permuted|permuted_both|permuted_both+reorder]   -> one JSON document on stdout
(order=...: only that numbering of the config-5 meshes — one kernel trace per order)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

orders = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("order=")] or None
what = [a for a in sys.argv[1:] if not a.startswith("order=")] or ["c5", "c3"]
dev = torch.device("cuda:0")
out = {}
if "c5" in what:
    sec = bench.c5_secondary(dev, 0, orders=orders)
    out["c5"] = {"mean_edge_span": sec["mean_edge_span"],
                 "dirac": [{k: p[k] for k in ("layout", "order", "product", "ms_median", "frac", "frac_actual")} for p in sec["products"]],
                 "laplacian": [{k: p[k] for k in ("order", "product", "kernel", "band", "ms_median", "frac")} for p in sec["laplacian"]],
                 "frac_min_packed_by_order": sec["frac_min_packed_by_order"], "laplacian_frac_min_by_order": sec["laplacian_frac_min_by_order"]}
if "c3" in what:
    out["c3"] = bench.c3_order_secondary(dev)
print(json.dumps(out, indent=1))
