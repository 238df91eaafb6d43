"""bench.py — headline benchmark of the Surface-Network hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N>1: one rank per GPU.  Either the caller starts the ranks (python -m torch.distributed.run --nproc-per-node N bench.py
--gpus N ...: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) or — when WORLD_SIZE is not set — bench.py
starts them itself through the same launcher on 127.0.0.1; rank 0 prints the one JSON line either way.

Workload (BASELINE.json configs[2], the config the metric "meshes/sec fwd+bwd, Dirac temporal-predict" is quoted on):
as_rigid_as_possible temporal prediction, Dirac model (15 blocks @128 ch, 1 018 872 params), 64 grid-cloth meshes of
71x71 vertices (V=5041, F=9800) per GPU, fp32, synthetic data, random-init weights.  A step = batch assembly on the
GPU + forward + masked smooth-L1 loss + backward + flat-bucket RCCL gradient all-reduce (N>1) + Adam update.
Weak scaling: every rank owns its own 64 meshes (the path shards by mesh; no data-path collective).

The line on stdout is COMPACT (compact_line: < 4 KB — the contract keys, config / roofline / cpu_baseline as scalars, one number
per secondary configuration); the FULL result described below goes to bench_detail.json next to this script (and to gpurun_out/
when present), path named in the line under "detail".

The full result carries
  roofline     the dominant SpMM kernel of the timed steps (the quaternion-packed Dirac product with the fused ELU-backward
               epilogue, N=32 dense columns), every launch timed live with HIP events that carry the kernel's own start/stop
               (hipExtLaunchKernelGGL) on the launch stream; achieved = ALGORITHMIC bytes (nnz*8 + (M+1)*4 +
               K*N*4 + M*N*4, plus the epilogue operands of the fused launches) / average launch duration; peak = 8 TB/s
               HBM3E; frac_by_convention reports the alternative readings; traffic = in-step PMC of the same command.
  cpu_baseline the reference's own CPU torch.sparse path (oracle restatement = "port") timed on this box's host cores
               (all of them, plus a 1-thread figure) on a bounded sample of the same workload (rank 0, N=1 only).
               config2_swap / config4_swap / config3_swap: configs[1] / [3] / [2] as the reference's own loops run them after the import
               swap (its batching names per step, eager), on meshes as generated and (`shuffled_ms_per_step`) with vertices and
               faces in random order.
  secondary    BASELINE.json configs[4] (the config north_star's ">= 60 % of the HBM roofline on the Dirac SpMM at 128
               channels" is quoted on): 128 meshes per GPU with 1 000 .. 20 000 vertices, Di / Di^T / DiA / DiA^T at N = 32,
               as a PACKED (unpadded, ragged) batch and — grid order only — padded to the batch maximum as the reference
               batches; algorithmic bytes always from the real sum of V_i, F_i (`frac`), next to the bytes the packed
               records really occupy (`actual_bytes`, `frac_actual`) and the HBM traffic of an in-run counter pass (`traffic`,
               `frac_traffic`).  Every entry names the vertex / face ORDER it was measured on (C5_ORDERS: the generator's grid
               order, SURVEY 8d's random vertex permutation, vertices and faces both shuffled, the latter stored in the
               product's locality numbering).  `laplacian`: the same meshes' cotangent Laplacians at 128 channels (L, L^T).
               `config3_order`: the headline step on shuffled meshes, stored as they came / renumbered.  `config3_swap`: the
               headline step behind the reference's own batching names and model calling sequence (an unmodified driver after
               the import swap).  `config2` / `config4_pair` / `config4_dp`: BASELINE.json configs[1] (Mesh-MNIST Dirac model,
               batch 512) and configs[3] (one pair of 6890-vertex bodies per rank, Laplacian towers): training steps replayed
               from a hipGraph, ms per step and meshes/s.
  roofline.linear_kernels  the Linear-layer kernels of the same timed steps (forward / input gradient / weight gradient of
               the folded BatchNorm+Linear: two thirds of the step), each launch timed the same way: launches, average
               duration, algorithmic bytes (operands read + results written) and TB/s.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
GRID = (71, 71)            # V = 5041, F = 9800
MESHES_PER_GPU = 64


def csrc_digest():
    """sha256 over the kernel sources and the ABI header: stamps profiles/*pmc_traffic*.json (tools/pmc_bench.py) so that a
    counter file measured on other kernels is not quoted for these."""
    import hashlib

    h = hashlib.sha256()
    src = os.path.join(ROOT, "surfacenetworks_amd", "csrc")
    for fn in sorted(os.listdir(src)) + [os.path.join("..", "..", "include", "sn_spmm.h")]:
        path = os.path.join(src, fn)
        if os.path.isfile(path) and (fn.endswith(".hip") or fn.endswith(".h")):
            h.update(os.path.basename(fn).encode())
            with open(path, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()


def alg_bytes(M, K, nnz, N, tag=""):
    """Algorithmic bytes of one product: CSR entries + row pointers + X + Y; a fused ELU-backward epilogue
    also reads the activation output E (+e) and the other branch's gradient G (+g), M x N floats each."""
    extra = (1 if "+e" in tag else 0) + (1 if "+g" in tag else 0)
    return nnz * 8 + (M + 1) * 4 + K * N * 4 + M * N * 4 * (1 + extra)


LINE_LIMIT = 4096          # bytes: the driver keeps a bounded tail of stdout; round 5's 25.6 KB line was not parsed
DETAIL_FILE = "bench_detail.json"


def _r(v, nd=4):
    """Scalars of the line: floats to nd significant digits after the point where it matters; everything else as is."""
    if isinstance(v, float):
        return float(f"{v:.{nd}g}") if abs(v) < 1 else round(v, 3)
    return v


def compact_line(full: dict, detail_path=None) -> dict:
    """The ONE line the driver parses: the contract keys, `config` / `roofline` / `cpu_baseline` reduced to scalars and short
    strings, `secondary` to one number per configuration and vertex order.  Everything else (per-product tables, the Linear
    kernels, the CPU legs, the long notes) is the FULL result, written to DETAIL_FILE next to this script (and copied under
    profiles/ by tools/round_profiles.sh); the line names the path.  tests/test_bench_line.py keeps it under LINE_LIMIT."""
    top = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
           "dtype", "data", "meshes_per_s")
    out = {k: _r(full[k]) for k in top if k in full}
    cfg = full.get("config") or {}
    keep = ("workload", "meshes_per_gpu", "global_batch", "global_pairs", "parallelism", "world_size", "rccl_ranks", "collective_backend",
            "devices_visible", "ranks_share_devices", "replicas_identical_after_the_run", "operator_format", "operators", "launch",
            "graph_fallback", "grad_bucket_bytes", "host_enqueue_ms_per_step", "host_enqueue_in_loop_ms_per_step", "host_affinity")
    out["config"] = {k: (_r(cfg[k]) if not isinstance(cfg[k], str) else cfg[k][:200]) for k in keep if cfg.get(k) is not None}
    roof = full.get("roofline")
    if roof:
        st = roof.get("step_traffic") or {}
        lin = roof.get("linear_kernels") or []
        out["roofline"] = {
            "bound": roof["bound"], "kernel": roof["kernel"].split(" (")[0], "achieved": _r(roof["achieved"]), "peak": roof["peak"],
            "unit": roof["unit"], "frac": _r(roof["frac"]), "frac_convention": roof.get("frac_convention", "").split(" (")[0].split(" /")[0],
            "frac_by_convention": {k: _r(v) for k, v in (roof.get("frac_by_convention") or {}).items()},
            "traffic": roof.get("traffic"), "algorithmic_bytes_per_launch": roof.get("algorithmic_bytes_per_launch"),
            "avg_launch_ms": _r(roof.get("avg_launch_ms")), "launches_timed": roof.get("launches_timed"),
            "step_traffic_GB": _r(st.get("GB_per_step")) if st else None,
            "spmm_ms_per_step": _r(roof.get("spmm_ms_per_step_all_kernels")), "linear_ms_per_step": _r(roof.get("linear_ms_per_step")),
            "linear_frac_min": _r(min((d["frac"] for d in lin), default=None)) if lin else None,
            "traffic_source": (roof.get("traffic_source") or "")[:60]}
    else:
        out["roofline"] = None
    cpu = full.get("cpu_baseline")
    if cpu:
        out["cpu_baseline"] = {"value": _r(cpu.get("value")), "unit": cpu.get("unit"), "cores": cpu.get("cores"), "kind": cpu.get("kind"),
                               "full_batch_value": _r(cpu.get("full_batch_value")), "sample": (cpu.get("sample") or "")[:300]}
    else:
        out["cpu_baseline"] = None
    sec = full.get("secondary")
    if sec:
        s = {}
        for k in ("frac_min_packed", "GBps_mean_packed", "aggregate_GBps_all_ranks_packed_mean"):
            if k in sec:
                s[k] = _r(sec[k])
        for k in ("frac_min_packed_by_order", "frac_actual_min_packed_by_order", "laplacian_frac_min_by_order"):
            if k in sec:
                s[k] = {o: _r(v) for o, v in sec[k].items()}
        for cfg_key in ("config4_dp", "config2", "config4_pair", "config3_swap", "config2_swap", "config4_swap"):
            c = sec.get(cfg_key)
            if isinstance(c, dict):
                s[cfg_key] = ({"error": c["error"][:120]} if "error" in c else
                              {k: _r(c[k]) for k in ("ms_per_step", "eager_ms_per_step", "host_enqueue_ms_per_step", "meshes_per_s",
                                                     "shuffled_ms_per_step", "renumbered_ms_per_step") if c.get(k) is not None})
        c = sec.get("config3_order")
        if isinstance(c, dict):
            s["config3_order_ms_per_step"] = ({"error": c["error"][:120]} if "error" in c else
                                              {o: _r(v["ms_per_step"]) for o, v in c.items() if isinstance(v, dict) and "ms_per_step" in v})
        if "error" in sec:
            s["error"] = sec["error"][:160]
        out["secondary"] = s
    out["detail"] = detail_path
    return out


def emit(full: dict, result_fd: int, detail_name: str = DETAIL_FILE):
    """Write the full result to <repo>/<detail_name> (and to gpurun_out/ when that directory exists: it travels back from the
    GPU box), then the compact line to the saved stdout descriptor."""
    path = None
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if d == ROOT or os.path.isdir(d):
            try:
                with open(os.path.join(d, detail_name), "w") as fh:
                    json.dump(full, fh, indent=1)
                path = path or os.path.relpath(os.path.join(d, detail_name), ROOT)
            except OSError:
                pass
    line = json.dumps(compact_line(full, path), separators=(",", ":"))
    if len(line) >= LINE_LIMIT:            # never again an unparseable record: drop the optional blocks, largest first
        slim = compact_line(full, path)
        for k in ("secondary",):
            slim[k] = {"dropped": f"line would be {len(line)} bytes; see {path}"}
        line = json.dumps(slim, separators=(",", ":"))
    sys.stdout.flush()
    os.write(result_fd, (line + "\n").encode())


def _cpu_leg(sample_meshes: int, seed: int, threads: int, budget_s: float, max_reps: int):
    """One fwd+bwd+Adam step loop of the reference path on `sample_meshes` meshes with `threads` threads: meshes/s."""
    from oracle import ref_blocks as OB          # checker/baseline only — never the measured product path
    from surfacenetworks_amd import mesh_ops

    torch.set_num_threads(threads)
    rng = np.random.default_rng(seed)
    per_mesh = []
    for _ in range(sample_meshes):
        V, F_ = mesh_ops.grid_cloth(*GRID, rng)
        Di, DiA = mesh_ops.dirac(V, F_)
        per_mesh.append((V.astype(np.float32), Di.astype(np.float32), DiA.astype(np.float32), F_.shape[0]))
    nv, nf = per_mesh[0][0].shape[0], per_mesh[0][3]
    model = OB.ArapDirModel().train()
    opt = torch.optim.Adam(model.parameters(), 1e-3, weight_decay=1e-5)
    inputs = torch.from_numpy(np.stack([np.concatenate([m[0], m[0]], 1) for m in per_mesh]))
    targets = torch.zeros(sample_meshes, nv, 120)
    mask = torch.ones(sample_meshes, nv, 1)

    def step():
        Di = OB.diag_cat([OB.sp_to_coo(m[1]) for m in per_mesh], 4 * nf, 4 * nv)      # per-step host batching, as the reference
        DiA = OB.diag_cat([OB.sp_to_coo(m[2]) for m in per_mesh], 4 * nv, 4 * nf)
        out = model(Di, DiA, mask, inputs)
        loss = OB.arap_loss(out, targets, mask, sample_meshes)
        opt.zero_grad()
        loss.backward()
        opt.step()

    if max_reps > 0:
        step()                               # warm-up (allocator, thread pool)
    else:
        max_reps = 1                         # (max_reps = 0: ONE cold step, no warm-up — the full-batch leg)
    t0 = time.perf_counter()
    reps = 0
    while reps < 1 or (time.perf_counter() - t0 < budget_s and reps < max_reps):
        step()
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    return sample_meshes / dt, reps


def _cpu_leg_subprocess(sample_meshes: int, seed: int, threads: int, budget_s: float, max_reps: int, limit_s: float):
    """Run one leg in its own process (own OpenMP pool, killable): {"value", "reps"} or {"timed_out": limit}."""
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-leg", f"{sample_meshes},{seed},{threads},{budget_s},{max_reps}"]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=limit_s)     # kills exactly this child on timeout
    except subprocess.TimeoutExpired:
        return {"threads": threads, "meshes": sample_meshes, "timed_out_after_s": limit_s}
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"threads": threads, "meshes": sample_meshes, "error": r.stderr[-300:]}
    return json.loads(lines[-1])


def cpu_baseline(seed: int):
    """The reference path (torch.mm(sparse_coo, dense) + autograd, oracle/ref_blocks.py) on the host cores, as SURVEY.md
    §8d asks: the same model, mesh shape and step (incl. the reference's per-step host batching sparse_diag_cat + coalesce)
    on a bounded sample, with torch.set_num_threads(os.cpu_count()) AND with 1 thread — plus one leg on the physical cores
    of one socket, because the all-logical-CPU setting the survey prescribes is pathological for this path on the 2 x 64-core
    SMT host (measured: 35x SLOWER than one thread — OpenMP barriers across 256 spinning threads around thousands of small
    sparse ops).  Every leg runs in its own time-boxed process.  `value` is the FASTEST leg (the fair baseline); every leg
    is listed under `legs`."""
    cores = os.cpu_count() or 1
    mid = max(1, min(64, cores // 4))
    legs = [_cpu_leg_subprocess(1, seed, 1, 6.0, 4, 60.0),
            _cpu_leg_subprocess(4, seed, mid, 8.0, 20, 60.0)]
    if cores not in (1, mid):
        legs.append(_cpu_leg_subprocess(1, seed, cores, 5.0, 2, 30.0))
    # the GPU workload's own configuration, once: ONE fwd+bwd+Adam step on all 64 meshes at the mid thread count (no warm-up
    # step: ~45-60 s on the 2 x 64-core host)
    full = _cpu_leg_subprocess(MESHES_PER_GPU, seed, mid, 1.0, 0, 240.0)
    full["note"] = "one cold step on the full per-GPU batch"
    legs.append(full)
    try:
        with open("/proc/cpuinfo") as fh:
            cpu = next((ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")), "unknown")
    except OSError:
        cpu = "unknown"
    done = [l_ for l_ in legs if "value" in l_]
    if not done:
        return {"value": None, "unit": "meshes/s", "cores": cores, "kind": "port", "sample": "no leg finished", "legs": legs}
    best = max(done, key=lambda l_: l_["value"])
    return {"value": best["value"], "unit": "meshes/s", "cores": best["threads"], "kind": "port",
            "full_batch_value": full.get("value"),
            "sample": f"{best['reps']} step(s) of fwd+bwd+Adam incl. per-step sparse_diag_cat on {best['meshes']} mesh(es) {GRID[0]}x{GRID[1]} "
                      f"(same model/shape as the GPU workload, 1/{MESHES_PER_GPU // best['meshes']} of the per-GPU batch), "
                      f"torch {torch.__version__} CPU torch.sparse path, {best['threads']} threads (fastest of the legs) on "
                      f"{cpu}, {cores} logical CPUs",
            "legs": legs}


# ---- secondary: BASELINE configs[4], the Dirac SpMM roofline batch ------------------------------------------------------
C5_MESHES_PER_GPU = 128
C5_VMIN, C5_VMAX = 1000, 20000


def _c5_meshes(rank: int, permute=False, reorder=False):
    """The 128 config-5 meshes of this rank (sizes from seed 5 + rank, the same in every variant): per-mesh Di, DiA, L and
    (sum V, sum F).  permute / reorder as arap.ClothSequences takes them."""
    from surfacenetworks_amd import mesh_ops

    rng = np.random.default_rng(5 + rank)
    vs = rng.integers(C5_VMIN, C5_VMAX + 1, size=C5_MESHES_PER_GPU)
    Dis, DiAs, Ls, sumV, sumF, spans = [], [], [], 0, 0, []
    for v in vs:
        n = int(np.sqrt(v))
        V, F_ = mesh_ops.grid_cloth(n, int(v) // n, rng, permute=permute)
        order = mesh_ops.MeshOrder.of_mesh(F_, V.shape[0], reorder)
        V, F_ = order.mesh(V, F_)
        spans.append(mesh_ops.edge_span(F_)[0])
        Di, DiA = mesh_ops.dirac(V, F_)
        Dis.append(Di.astype(np.float32))
        DiAs.append(DiA.astype(np.float32))
        Ls.append(mesh_ops.laplacian(V, F_).astype(np.float32))
        sumV += V.shape[0]
        sumF += F_.shape[0]
    return Dis, DiAs, Ls, sumV, sumF, float(np.mean(spans))


def _c5_time(o, x, y, group, iters, warm):
    from surfacenetworks_amd import functional as snF

    for _ in range(warm):
        snF._launch(o, x, y, group, "c5")
    timer = snF.SpmmTimer()
    with timer:
        for _ in range(iters):
            snF._launch(o, x, y, group, "c5")
    recs = timer.results()
    return np.array([r[5] for r in recs]), recs[0][0].split("/")[-1]


def _c5_dirac_products(pools, layouts, order, g, device, iters, warm, N=32):
    sel = np.arange(C5_MESHES_PER_GPU)
    out = []
    for layout in layouts:
        for name in ("Di", "DiA"):
            pool = pools[name]
            if layout == "packed":
                op = pool.assemble(sel)
            else:
                op = pool.assemble(sel, int(pool.rows.max()), int(pool.cols.max()))
            for prod, o in ((name, op), (name + "^T", op.t())):
                M, K = o.shape
                real_M = int((pool.rows if prod == name else pool.cols).sum())
                real_K = int((pool.cols if prod == name else pool.rows).sum())
                x = torch.randn(K // 4, 4 * N, device=device, generator=g)
                y = torch.empty(M // 4, 4 * N, device=device)
                ms, _ = _c5_time(o, x, y, 4, iters, warm)
                ab = alg_bytes(real_M, real_K, o.nnz, N)
                # what the packed form really moves: 16-byte quaternion records + block-row pointers + X + Y (real rows only:
                # the padded layout also WRITES its padding rows — charged to its time, not credited)
                q = o.q3()
                actual = (int(q[1].shape[0]) * 16 + (M // 4 + 1) * 4 + real_K * N * 4 + real_M * N * 4) if q is not None else None
                med = float(np.median(ms)) * 1e-3
                out.append({
                    "layout": layout, "order": order, "product": prod, "M": M, "K": K, "real_M": real_M, "real_K": real_K, "nnz": o.nnz,
                    "algorithmic_bytes": ab, "actual_bytes": actual, "ms_median": float(np.median(ms)), "ms_min": float(ms.min()),
                    "ms_mean": float(ms.mean()), "GBps": ab / med / 1e9, "frac": ab / med / HBM_PEAK,
                    "frac_actual": (actual / med / HBM_PEAK) if actual else None})
                del x, y
            del op
    return out


def _c5_laplacian_products(Ls, order, g, device, iters, warm):
    from surfacenetworks_amd.operators import OperatorPool

    pool = OperatorPool(Ls, device)
    op = pool.assemble(np.arange(C5_MESHES_PER_GPU))
    out = []
    for prod, o in (("L", op), ("L^T", op.t())):
        M, K = o.shape
        x = torch.randn(K, 128, device=device, generator=g)
        y = torch.empty(M, 128, device=device)
        ms, kernel = _c5_time(o, x, y, 1, iters, warm)
        ab = alg_bytes(M, K, o.nnz, 128)
        out.append({"layout": "packed", "order": order, "product": prod, "kernel": kernel, "band": list(o.band()),
                    "M": M, "K": K, "nnz": o.nnz, "algorithmic_bytes": ab,
                    "ms_median": float(np.median(ms)), "ms_min": float(ms.min()),
                    "GBps": ab / (float(np.median(ms)) * 1e-3) / 1e9,
                    "frac": ab / (float(np.median(ms)) * 1e-3) / HBM_PEAK})
        del x, y
    return out


# vertex / face numberings the config-5 products are measured on ("row-major grid, plus a random-permuted
# variant (seeded) to expose gather locality"): tag -> (permute, reorder) of _c5_meshes
C5_ORDERS = {
    "grid": (False, False),                           # the generator's row-major numbering
    "permuted": ("vertices", False),                  # §8d's variant: vertices renumbered at random, face list as generated
    "permuted_both": ("both", False),                 # faces shuffled too (a scanned mesh), dataset order kept as stored
    "permuted_both+reorder": ("both", True),          # the same meshes stored in the product's locality numbering (MeshOrder)
}


def c5_secondary(device, rank: int, iters: int = 50, warm: int = 10, orders=None):
    """Config 5: 128 grid-cloth meshes per GPU, V_i uniform in [1000, 20000] (seed 5 + rank), Dirac
    operators, N = 32 (C = 128).  The four products Di, Di^T, DiA, DiA^T are launched back to back (>= 50 timed launches
    after >= 10 warm-ups, the kernel's own start/stop in HIP events on the launch stream) on
      packed  the ragged batch without padding (OperatorPool.assemble(sel): prefix-sum offsets, operands (sum V_i, C));
      padded  every mesh padded to the batch maximum as the reference's sparse_diag_cat does (utils_pt.py:41-53).
    Algorithmic bytes = nnz*8 + (M+1)*4 + K*N*4 + M*N*4 with M, K from the REAL sum of 4*F_i / 4*V_i in both cases, so the
    padded variant gets no credit for the zeros it writes.  Every entry names the vertex / face ORDER it was measured on
    (C5_ORDERS): the generator's grid order, §8d's random vertex permutation, vertices and faces both shuffled, and the latter
    stored in the product's locality numbering."""
    from surfacenetworks_amd.operators import OperatorPool

    N = 32
    g = torch.Generator(device=device).manual_seed(7)
    out = None
    for order, (permute, reorder) in C5_ORDERS.items():
        if orders is not None and order not in orders:
            continue
        Dis, DiAs, Ls, sumV, sumF, span = _c5_meshes(rank, permute, reorder)
        if out is None:
            out = {"workload": f"BASELINE configs[4]: {C5_MESHES_PER_GPU} grid-cloth meshes per GPU, V in [{C5_VMIN}, {C5_VMAX}] "
                               f"(sum V = {sumV}, sum F = {sumF}), Dirac operators, C = 128 (N = {N}), quaternion-packed records",
                   "timing": f"{iters} back-to-back launches after {warm} warm-ups; hipExtLaunchKernelGGL start/stop events per launch",
                   "orders": {"grid": "row-major grid numbering (generator order)",
                              "permuted": "seeded random vertex numbering, face list as generated (SURVEY.md §8d), seed 5 + rank",
                              "permuted_both": "vertices renumbered and faces shuffled at random, used as stored",
                              "permuted_both+reorder": "the same shuffled meshes stored in mesh_ops.MeshOrder's locality numbering "
                                                       "(reverse Cuthill-McKee, one-time at pool construction)"},
                   "mean_edge_span": {}, "peak_GBps": HBM_PEAK / 1e9, "products": [], "laplacian": []}
        out["mean_edge_span"][order] = span
        pools = {"Di": OperatorPool(Dis, device, want_bsr4=True), "DiA": OperatorPool(DiAs, device, want_bsr4=True)}
        out["products"] += _c5_dirac_products(pools, ("packed", "padded") if order == "grid" else ("packed",), order, g, device, iters, warm, N)
        del pools
        # the same meshes' cotangent Laplacians at 128 channels, packed (the operator of the Laplacian models; default: the
        # sliding-window kernel straight from the CSR arrays where the band allows, else the row-blocked form) — reported
        # next to the Dirac products, not part of frac_min_packed (north_star's bar names the Dirac)
        out["laplacian"] += _c5_laplacian_products(Ls, order, g, device, iters, warm)
        del Dis, DiAs, Ls
        torch.cuda.empty_cache()
    done = [o for o in C5_ORDERS if orders is None or o in orders]
    packed = [p_ for p_ in out["products"] if p_["layout"] == "packed" and p_["order"] == done[0]]      # (done[0] = "grid" by default)
    out["frac_min_packed"] = min(p_["frac"] for p_ in packed)
    out["GBps_mean_packed"] = float(np.mean([p_["GBps"] for p_ in packed]))
    out["frac_min_packed_by_order"] = {o: min(p_["frac"] for p_ in out["products"] if p_["layout"] == "packed" and p_["order"] == o)
                                       for o in done}
    out["frac_actual_min_packed_by_order"] = {o: min((p_["frac_actual"] or p_["frac"]) for p_ in out["products"]
                                                     if p_["layout"] == "packed" and p_["order"] == o) for o in done}
    out["laplacian_frac_min_by_order"] = {o: min(p_["frac"] for p_ in out["laplacian"] if p_["order"] == o) for o in done}
    return out


def c5_pmc_child(plan_path: str, launches: int = 6):
    """Child of c5_pmc_traffic (runs under rocprofv3 --pmc): the packed config-5 products of every order, `launches` launches
    each, and the PLAN — (order, product, launches) in launch order — written to `plan_path`, so that the parent can attribute
    the profiler's dispatch records (the k-th run of sparse-product dispatches belongs to the k-th plan entry)."""
    from surfacenetworks_amd import functional as snF
    from surfacenetworks_amd.operators import OperatorPool

    device = torch.device("cuda:0")
    g = torch.Generator(device=device).manual_seed(7)
    sel = np.arange(C5_MESHES_PER_GPU)
    plan = []
    for order, (permute, reorder) in C5_ORDERS.items():
        Dis, DiAs, Ls, _, _, _ = _c5_meshes(0, permute, reorder)
        for name, mats, group, N in (("Di", Dis, 4, 32), ("DiA", DiAs, 4, 32), ("L", Ls, 1, 128)):
            op = OperatorPool(mats, device, want_bsr4=(group == 4)).assemble(sel)
            for prod, o in ((name, op), (name + "^T", op.t())):
                M, K = o.shape
                x = torch.randn(K // group, group * N, device=device, generator=g)
                y = torch.empty(M // group, group * N, device=device)
                if group == 1:
                    o.ring_ok(N), o.rb4() if not o.ring_ok(N) else None        # (derived forms built before the counted launches)
                torch.cuda.synchronize()
                for _ in range(launches):
                    snF._launch(o, x, y, group, "c5")
                torch.cuda.synchronize()
                plan.append({"order": order, "product": prod, "launches": launches})
            del op
        torch.cuda.empty_cache()
    with open(plan_path, "w") as fh:
        json.dump(plan, fh)


def c5_pmc_traffic(launches: int = 6):
    """HBM traffic per launch of the packed config-5 products, every order: this file re-executed (--c5-pmc-child) under
    rocprofv3, one counter per pass (bytes = RDREQ*128 + WRREQ*64, MI355X_MICROARCH.md); returns {(order, product): bytes}
    and a note, or (None, reason)."""
    import csv
    import glob
    import shutil
    import tempfile

    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="sn_pmc5_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    try:
        per = {}
        for ctr in (PMC_RD, PMC_WR):
            out, plan_path = os.path.join(tmp, ctr), os.path.join(tmp, ctr + "_plan.json")
            r = subprocess.run([exe, "--pmc", ctr, "-d", out, "-o", "pmc", "--output-format", "csv", "--", sys.executable,
                                os.path.abspath(__file__), "--c5-pmc-child", f"{plan_path},{launches}"], cwd="/tmp", env=env,
                               capture_output=True, text=True, timeout=600)
            hits = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not hits or not os.path.exists(plan_path):
                return None, f"rocprofv3 --pmc {ctr} over the config-5 products failed (rc {r.returncode}): {r.stderr[-160:]!r}"
            with open(plan_path) as fh:
                plan = json.load(fh)
            rows = sorted((int(x["Dispatch_Id"]), x["Kernel_Name"], float(x["Counter_Value"])) for x in csv.DictReader(open(hits[0]))
                          if x["Counter_Name"] == ctr)
            prods = [v for _, name, v in rows if "spmm_" in name and "stats_reduce" not in name]
            if len(prods) != sum(e["launches"] for e in plan):
                return None, f"{len(prods)} sparse-product dispatches recorded, {sum(e['launches'] for e in plan)} planned"
            k = 0
            for e in plan:
                vals = prods[k:k + e["launches"]]
                k += e["launches"]
                per.setdefault((e["order"], e["product"]), {})[ctr] = float(np.mean(vals[1:] if len(vals) > 1 else vals))   # (first launch: cold)
        traffic = {key: v[PMC_RD] * 128 + v[PMC_WR] * 64 for key, v in per.items()}
        return traffic, (f"in-run: bench.py --c5-pmc-child under rocprofv3 --pmc {PMC_RD} / {PMC_WR} (one counter per pass), mean of "
                         f"{launches - 1} launches per product; bytes = RDREQ*128 + WRREQ*64")
    except (subprocess.TimeoutExpired, OSError, ValueError, KeyError) as exc:
        return None, f"config-5 PMC pass failed: {exc!r}"[:200]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def c3_order_secondary(device, meshes: int = MESHES_PER_GPU, steps: int = 10, warm: int = 4):
    """The headline step (config 3) on meshes that do NOT arrive in grid order: 64 grid-cloth meshes 71x71 with vertices and
    faces shuffled (seed 3), trained (a) as stored and (b) stored in the product's locality numbering (ClothSequences(reorder=
    True): one-time host work at dataset construction, nothing per step).  Eager steps, same step definition as the headline."""
    from surfacenetworks_amd import arap

    out = {"workload": f"config 3 with shuffled vertex and face numbering: {meshes} meshes {GRID[0]}x{GRID[1]}, Dirac model, "
                       "assembly + fwd + loss + bwd + Adam, eager", "steps": steps, "warmup": warm}
    for tag, permute, reorder in (("permuted_both", "both", False), ("permuted_both+reorder", "both", True), ("permuted", "vertices", False)):
        ds = arap.ClothSequences([GRID] * meshes, frames=arap.INPUT_FRAMES + arap.OUTPUT_FRAMES + 2, op_frames=2, seed=3,
                                 device=device, model="dir", permute=permute, reorder=reorder)
        torch.manual_seed(1234)
        model = arap.DirModel().to(device).train()
        opt = arap.make_optimizer(model)
        rng = np.random.default_rng(10)
        ids = np.arange(meshes)
        dt = _timed_steps(lambda: arap.train_step(model, opt, ds.sample_batch(meshes, rng, seq_ids=ids)), steps, warm)
        out[tag] = {"ms_per_step": dt * 1e3, "meshes_per_s": meshes / dt}
        del ds, model, opt
        torch.cuda.empty_cache()
    return out


def _swap_bench(name, device, **kw):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import train_bench

    from surfacenetworks_amd.resident import resident_cache

    out = getattr(train_bench, name)(str(device), **kw)
    torch.cuda.empty_cache()
    c = resident_cache()
    if c is not None:
        c.clear()                                   # (the shuffled twin below brings its own matrices)
    shuffled = getattr(train_bench, name)(str(device), permute=True, **kw)
    out["shuffled_ms_per_step"] = shuffled["ms_per_step"]
    if c is not None:
        c.clear()
    return out


def c3_swap_secondary(device, steps: int = 10):
    """Config 3 behind the reference's OWN names (north_star: the training scripts "run unmodified apart from an import swap"):
    tools/train_bench.py arap_swap — per step sp_sparse_to_pt_sparse per sample, sparse_diag_cat, .cuda(), the reference's model
    calling sequence and loss — next to the headline, which uses the product's own sampler (ClothSequences / OperatorPool).
    `shuffled_ms_per_step`: the same loop on meshes whose vertices and faces arrive in random order, multiplied as stored (the
    resident cache packs matrices as given: a scanned mesh in file order)."""
    return _swap_bench("arap_swap", device, steps=steps, B=MESHES_PER_GPU)


def c2_swap_secondary(device, steps: int = 30):
    """Config 2 as the reference's Mesh-MNIST loop runs it after the import swap (tools/train_bench.py mnist_swap), eager."""
    return _swap_bench("mnist_swap", device, steps=steps)


def c4_swap_secondary(device, steps: int = 30):
    """Config 4 (one pair) as the reference's dense-correspondence loop runs it after the import swap (tools/train_bench.py
    faust_swap), eager."""
    return _swap_bench("faust_swap", device, steps=steps)


def _timed_steps(step, steps, warm):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def c2_secondary(device, steps: int = 60, warm: int = 10):
    """BASELINE configs[1]: Mesh-MNIST Dirac model (5 Dirac blocks at 64 channels, src/mesh_mnist/models.py:122-159), batch 512 of
    ~150-vertex meshes, fp32; a step = batch assembly + forward + NLL + backward + Adam (src/mesh_mnist/main.py:151-167),
    forward+loss+backward replayed from one hipGraph (the step is ~250 launches of a few microseconds)."""
    from surfacenetworks_amd import mesh_mnist as mm

    B = 512
    rng = np.random.default_rng(2)
    ds = mm.MeshDigits(B, seed=2, device=device, fixed_vertices=150, model="dir")
    model = mm.DirModel().to(device).train()
    opt = mm.make_optimizer(model)
    ids = np.arange(B)
    eager = _timed_steps(lambda: mm.train_step(model, opt, ds.sample_batch(B, rng, ids=ids)), 10, 14)   # (steady state: plan graphs made)
    from surfacenetworks_amd.graphs import BatchAhead

    g = mm.graphed_train_step(model, opt, ds.sample_batch(B, rng, ids=ids))
    serial = _timed_steps(lambda: g(ds.sample_batch(B, rng, ids=ids)), steps, warm)
    ahead = BatchAhead(lambda: ds.sample_batch(B, rng, ids=ids), device)       # batch t+1 assembled while step t computes
    dt = _timed_steps(lambda: g(ahead.get()), steps, warm)
    return {"workload": "BASELINE configs[1]: Mesh-MNIST Dirac model, batch 512, 150-vertex meshes, C = 64, fp32; batch assembly + "
                        "fwd + NLL + bwd + Adam",
            "launch": "hipGraph replay of fwd+loss+bwd; sampling (one batch ahead, on a side stream) and Adam eager",
            "steps": steps, "warmup": warm, "ms_per_step": dt * 1e3, "meshes_per_s": B / dt,
            "ms_per_step_assembly_on_the_compute_stream": serial * 1e3, "eager_ms_per_step": eager * 1e3}


def c4_pair_secondary(device, steps: int = 60, warm: int = 10):
    """The per-GPU work of BASELINE configs[3] (FAUST dense correspondence, 8 GPUs data parallel, one pair per GPU and step:
    src/dense_correspondence/main.py:40,310-327): two 6890-vertex bodies padded to 7000 vertices, Laplacian towers (15 blocks at
    128 channels), the 7000 x 7000 score matrix, argmin-target cross entropy, backward, Adam; replayed from one hipGraph."""
    from surfacenetworks_amd import dense_correspondence as dc

    ds = dc.TorusBodies(4, device=device)
    model = dc.SiameseModel("lap", 15).to(device).train()
    opt = dc.make_optimizer(model)
    k = [0]

    def estep():
        k[0] += 1
        dc.train_step(model, opt, ds, k[0] % 4, (k[0] + 1) % 4)
    eager = _timed_steps(estep, 12, 52)              # (steady state: every address set of the four-pair cycle has its plan graphs)
    g = dc.graphed_train_step(model, opt, dc.PairBatch(ds, 0, 1))

    def gstep():
        k[0] += 1
        g(dc.PairBatch(ds, k[0] % 4, (k[0] + 1) % 4))
    dt = _timed_steps(gstep, steps, warm)
    return {"workload": "per-GPU work of BASELINE configs[3]: one FAUST-sized pair (2 x 6890 vertices padded to 7000), Laplacian "
                        "towers C = 128, 15 blocks, 7000 x 7000 score matrix + argmin-target cross entropy, bwd, Adam",
            "launch": "hipGraph replay of fwd+loss+bwd; pair selection and Adam eager",
            "steps": steps, "warmup": warm, "ms_per_step": dt * 1e3, "meshes_per_s": 2 / dt, "pairs_per_s": 1 / dt,
            "eager_ms_per_step": eager * 1e3}


def c4_dp_step(device, rank: int, world: int, steps: int = 40, warm: int = 8):
    """BASELINE configs[3] as the N-rank job it is (src/dense_correspondence/main.py:40,299-327: batch 1 per device): every
    rank trains on ITS OWN pair of FAUST-sized bodies per step — forward + loss + backward replayed from one hipGraph —, then
    one flat-bucket all-reduce (SUM; the loss is normalised by the number of pairs in the job) and Adam.  Returns the local
    seconds per step and the bookkeeping of the run; the caller takes the MAX over ranks."""
    import torch.distributed as dist

    from surfacenetworks_amd import dense_correspondence as dc
    from surfacenetworks_amd import dp

    ds = dc.TorusBodies(4, device=device, seed=4 + 1000 * rank)
    torch.manual_seed(4321)
    model = dc.SiameseModel("lap", 15).to(device).train()
    dp.broadcast_parameters(model, 0)
    opt = dc.make_optimizer(model)
    bucket = dp.FlatGradBucket(model.parameters(), always_reduce=dist.is_initialized())
    k = [0]
    launch = "hipGraph replay of fwd+loss+bwd; pair selection, all-reduce, Adam eager"
    try:
        g = dc.graphed_train_step(model, opt, dc.PairBatch(ds, 0, 1), bucket=bucket, global_pairs=world)

        def step():
            k[0] += 1
            return g(dc.PairBatch(ds, k[0] % 4, (k[0] + 1) % 4))
    except Exception as exc:  # noqa: BLE001  (a rank whose capture fails runs the same step eagerly: same collective per step)
        launch = f"eager (graph capture failed: {type(exc).__name__}: {exc})"[:300]
        try:
            torch.cuda.synchronize()
        except Exception:  # noqa: BLE001
            pass

        def step():
            k[0] += 1
            return dc.train_step(model, opt, ds, k[0] % 4, (k[0] + 1) % 4, grad_sync=bucket.sync, global_pairs=world,
                                 zero_grads=bucket.detach_grads)

    for _ in range(warm):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    t_enq = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert torch.isfinite(loss.detach()).item(), "FAUST step diverged"
    return dt / steps, {"steps": steps, "warmup": warm, "host_enqueue_ms_per_step": t_enq / steps * 1e3,
                        "grad_bucket_bytes": bucket.nbytes, "launch": launch,
                        "replicas_identical_after_the_run": replicas_agree(model, world)}


def gpu_numa_node(device_index: int, sysfs: str = "/sys") -> int:
    """NUMA node of a HIP device from its PCI address (hipDeviceGetPCIBusId -> <sysfs>/bus/pci/devices/<bdf>/numa_node);
    -1 when the platform does not say (single-node hosts, containers, a missing sysfs entry, no HIP runtime)."""
    try:
        import ctypes

        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(device_index)) != 0:
            return -1
        with open(os.path.join(sysfs, "bus/pci/devices", buf.value.decode().lower(), "numa_node")) as fh:
            return int(fh.read().strip())
    except (OSError, ValueError, AttributeError):
        return -1


def cpus_for_rank(node: int, cpus_all, local_rank: int, n_local: int, sysfs: str = "/sys"):
    """(cpus, description): the cores of NUMA node `node` that this process may use; without a node (-1), or when its cpulist
    is missing / disjoint from the allowed set, the allowed CPUs are dealt out evenly among the node's ranks (a rank beyond the
    last CPU shares them all)."""
    cpus_all = sorted(cpus_all)
    cpus = None
    if node >= 0:
        try:
            with open(os.path.join(sysfs, f"devices/system/node/node{node}/cpulist")) as fh:
                want = set()
                for part in fh.read().strip().split(","):
                    a, _, b = part.partition("-")
                    want.update(range(int(a), int(b or a) + 1))
            cpus = [c for c in cpus_all if c in want]
        except (OSError, ValueError):
            cpus = None
    if cpus:
        return cpus, f"numa node {node}: {len(cpus)} cpus"           # (the ranks of one NUMA node share its cores)
    per = max(1, len(cpus_all) // max(1, n_local))
    share = cpus_all[local_rank * per:(local_rank + 1) * per] or cpus_all
    return share, f"no numa node for the device: cpus {share[0]}-{share[-1]} ({len(share)}) of {len(cpus_all)}"


def pin_to_gpu_numa(local_rank: int, n_local: int):
    """Keep this rank's host threads on the cores next to its GPU: eight eager-launching Python processes on a two-socket
    host otherwise migrate across sockets (gpu_numa_node / cpus_for_rank).  Returns a short description for the JSON line."""
    try:
        share, how = cpus_for_rank(gpu_numa_node(torch.cuda.current_device()), os.sched_getaffinity(0), local_rank, n_local)
        os.sched_setaffinity(0, share)
        torch.set_num_threads(max(1, min(8, len(share))))
        return how
    except Exception as exc:  # noqa: BLE001 — affinity is an optimisation, never a reason to fail the run
        return f"not pinned: {exc!r}"[:120]


PMC_RD, PMC_WR = "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum"


def pmc_in_run(args, steps: int = 4):
    """HBM traffic of THIS run's kernels: re-execute this file for a few steps of the same workload under rocprofv3, one
    counter per pass (MI355X_MICROARCH.md: requests counted at the L2's memory side; bytes = RDREQ*128 + WRREQ*64 on gfx950),
    and condense the per-dispatch counters with tools/pmc_bench.py.  Returns (table, source note); (None, reason) when
    rocprofv3 is missing or a pass fails — the caller then falls back to the stamped file under profiles/."""
    import glob
    import shutil
    import tempfile

    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="sn_pmc_", dir="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--steps", str(steps), "--warmup", "1", "--no-cpu-baseline", "--no-secondary",
             "--no-pmc", "--backend", "none", "--meshes", str(args.meshes), "--format", args.format, "--operators", args.operators,
             "--linear-timing-steps", "0"]
    env = dict(os.environ, TMPDIR="/tmp")
    env.pop("RANK", None), env.pop("WORLD_SIZE", None), env.pop("LOCAL_RANK", None)
    found = {}
    try:
        for ctr in (PMC_RD, PMC_WR):
            out = os.path.join(tmp, ctr)
            r = subprocess.run([exe, "--pmc", ctr, "-d", out, "-o", "pmc", "--output-format", "csv", "--", *child], cwd="/tmp", env=env,
                               capture_output=True, text=True, timeout=420)
            hits = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not hits:
                return None, f"rocprofv3 --pmc {ctr} failed (rc {r.returncode}): {r.stderr[-160:]!r}"
            found[ctr] = hits[0]
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import pmc_bench

        # the child runs data set-up, 3 untimed eager steps, 1 warm-up and `steps` timed steps: keep exactly the timed steps
        res, _agg, _rd, _wr, _n0, tot_r, tot_w = pmc_bench.summarize(found[PMC_RD], found[PMC_WR], last_steps=steps)
        res["_steps_counted"] = steps
        return res, (f"in-run: this bench invocation re-executed itself for {steps} steps under rocprofv3 --pmc {PMC_RD} / --pmc {PMC_WR} "
                     f"(one counter per pass); bytes = RDREQ*128 + WRREQ*64")
    except (subprocess.TimeoutExpired, OSError, ValueError, KeyError, ImportError) as exc:
        return None, f"in-run PMC pass failed: {exc!r}"[:200]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def replicas_agree(model, world: int) -> bool:
    """After the timed steps every rank must hold the same parameters, bit for bit (same initial broadcast, same all-reduced
    gradients, same Adam): compared through an all-gather of two checksums per rank."""
    import torch.distributed as dist

    flat = torch.cat([p.detach().reshape(-1).double() for p in model.parameters()])
    mine = torch.stack([flat.sum(), (flat * torch.arange(1, flat.numel() + 1, device=flat.device, dtype=torch.float64)).sum()])
    if world == 1 or not dist.is_initialized():
        return True
    got = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(got, mine)
    return bool(all(torch.equal(g, got[0]) for g in got))


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start the N ranks here, through the same
    launcher the contract names (python -m torch.distributed.run, rendezvous on 127.0.0.1).  With fewer visible GPUs than
    ranks (a 1-GPU box) the ranks share devices and the collective backend falls back to gloo — a functional run of the
    N > 1 path, flagged as such in the JSON line."""
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(args.gpus, 1))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--meshes", type=int, default=MESHES_PER_GPU, help="meshes per GPU (default: the BASELINE config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="replay forward+loss+backward from one captured hipGraph per step instead of launching every "
                         "kernel from Python (same kernels; pays off when the step is launch-bound: small batches, busy hosts). "
                         "Default with more than one rank (N eager-launching Python processes share one host); --eager overrides")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from Python also with more than one rank")
    ap.add_argument("--workload", default="arap", choices=["arap", "faust"],
                    help="arap (default): BASELINE configs[2], the headline; faust: BASELINE configs[3] — dense correspondence, one "
                         "pair of 6890-vertex bodies per rank and step, Laplacian towers, hipGraph replay + flat-bucket all-reduce")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the in-run counter passes (rocprofv3 --pmc over a few steps of this same workload, N = 1 only)")
    ap.add_argument("--roofline-steps", type=int, default=5,
                    help="eager steps run AFTER the timed region with per-launch HIP events (graph mode only)")
    ap.add_argument("--linear-timing-steps", type=int, default=5,
                    help="timed steps whose Linear-layer launches carry start/stop events as well (roofline.linear_kernels).  An event "
                         "pair costs a launch ~2 us: 0.3 ms on all of a step's launches")
    ap.add_argument("--spmm-timing-steps", type=int, default=5,
                    help="timed steps (the first ones of the timed region) whose sparse products carry start/stop events (roofline."
                         "achieved): 32 launches per step, 0.10-0.15 ms per step when carried on all 20 (same box, three reps: 19.17 / "
                         "19.24 / 19.30 ms against 19.08 / 19.11 / 19.15 with 5); 0 or less: every timed step")
    ap.add_argument("--format", default="q3", choices=["q3", "bsr4", "csr"],
                    help="Dirac operator form: quaternion-packed blocks (default), 4x4 blocks, generic CSR")
    ap.add_argument("--operators", default="pool", choices=["pool", "device"],
                    help="pool: precomputed per-frame operators resident in HBM (default, = the reference's dataset); "
                         "device: Dirac operators rebuilt on the GPU from the frame coordinates every step")
    ap.add_argument("--backend", default=None, choices=[None, "nccl", "gloo", "none"],
                    help="default nccl (= RCCL) — also with ONE rank: a one-rank communicator, the gradient bucket goes "
                         "through ncclAllReduce every step; gloo only for functional tests of the N>1 path on a 1-GPU box "
                         "(chosen automatically when there are fewer visible GPUs than ranks); none: no process group at N = 1")
    ap.add_argument("--no-secondary", action="store_true", help="skip the config-5 SpMM roofline block")
    ap.add_argument("--cpu-leg", default=None, help=argparse.SUPPRESS)      # internal: one time-boxed leg of cpu_baseline()
    ap.add_argument("--c5-pmc-child", default=None, help=argparse.SUPPRESS)  # internal: the config-5 products under the profiler
    args = ap.parse_args()
    world_hint = int(os.environ.get("WORLD_SIZE", str(args.gpus)))
    # N ranks on N devices: replay unless --eager (N eager-launching Python processes share one host).  Ranks that SHARE a
    # device (the functional run of the N > 1 path on a 1-GPU box) stay eager: replaying two processes' graphs on one device
    # measured 1.4 s per step against 42 ms eager
    try:
        n_dev = torch.cuda.device_count()
    except Exception:  # noqa: BLE001
        n_dev = 0
    args.graph_default = (world_hint > 1 and not args.eager and n_dev >= world_hint)
    args.no_graph = not (args.graph or args.graph_default)
    if args.cpu_leg:
        n_, seed_, thr_, bud_, reps_ = args.cpu_leg.split(",")
        v, reps = _cpu_leg(int(n_), int(seed_), int(thr_), float(bud_), int(reps_))
        print(json.dumps({"value": v, "reps": reps, "threads": int(thr_), "meshes": int(n_)}), flush=True)
        return
    if args.c5_pmc_child:
        plan_path, n_ = args.c5_pmc_child.rsplit(",", 1)
        c5_pmc_child(plan_path, int(n_))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args, sys.argv[1:]))

    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a version banner through C stdio when
    # its communicator goes away — after our line, once a one-rank group is part of every run): from here on file
    # descriptor 1 is stderr for everybody, and the line goes to the saved descriptor.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch.distributed as dist

    from surfacenetworks_amd import arap, dp, plans
    from surfacenetworks_amd import functional as snF

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path in the product)")
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    oversubscribed = world_env > torch.cuda.device_count()
    backend = args.backend or ("gloo" if oversubscribed else "nccl")    # RCCL cannot put two ranks on one device
    group_note = None
    if world_env == 1 and backend != "none":
        # one rank: still a process group, so that the step's gradient reduction is a real collective call
        if "MASTER_PORT" not in os.environ:
            sock = socket.socket()
            sock.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
            sock.close()
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        try:
            rank, local_rank, world, device = dp.init_distributed(backend, single_rank_group=True)
        except Exception as exc:  # noqa: BLE001 — the headline number must not depend on the one-rank communicator coming up
            group_note = f"one-rank {backend} group failed to initialise: {exc!r}"[:200]
            rank, local_rank, world, device = dp.init_distributed(None)
    else:
        rank, local_rank, world, device = dp.init_distributed(None if backend == "none" else backend)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    n_local_ranks = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    affinity = pin_to_gpu_numa(local_rank, n_local_ranks) if world > 1 else "single rank: not pinned"
    snF.set_dirac_format(args.format)
    torch.manual_seed(1234)
    if args.workload == "faust":
        # BASELINE configs[3] as its own line: N ranks, one pair per rank and step (same contract: barrier + synchronize on both
        # sides inside c4_dp_step, MAX over ranks, rank 0 prints)
        steps_f, warm_f = args.steps, args.warmup
        dt_f, info = c4_dp_step(device, rank, world, steps_f, warm_f)
        t = torch.tensor([dt_f], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_f = float(t.item())
        line = {"metric": "pairs/sec fwd+bwd, FAUST dense correspondence (BASELINE configs[3])", "value": world / dt_f, "unit": "pairs/s",
                "n_gpus": world, "steps": steps_f, "warmup": warm_f, "ms_per_step": dt_f * 1e3, "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "meshes_per_s": 2 * world / dt_f,
                "config": {"workload": "FAUST dense correspondence: one pair of torus-grid bodies (6890 vertices padded to 7000) per rank "
                                       "and step, Laplacian towers C=128, 15 blocks, score matrix never materialised, delta cross entropy, "
                                       "hipGraph replay of fwd+loss+bwd, pack + all-reduce(SUM) of the flat gradient bucket, Adam",
                           "global_pairs": world, "parallelism": f"dp{world} (one pair per rank, flat-bucket all-reduce)",
                           "world_size": world, "rccl_ranks": world if (dist.is_initialized() and dist.get_backend() == "nccl") else 0,
                           "collective_backend": dist.get_backend() if dist.is_initialized() else "none",
                           "devices_visible": torch.cuda.device_count(), "ranks_share_devices": bool(oversubscribed),
                           "host_affinity": affinity, **info},
                "roofline": None, "cpu_baseline": None}
        if rank == 0:
            emit(line, result_fd, "bench_detail_faust.json")
        if dist.is_initialized():
            if world > 1:
                dist.barrier()
            dist.destroy_process_group()
        return

    # ---- data: this rank's shard (own meshes; weak scaling) -----------------------------------------
    n_local = args.meshes
    ds = arap.ClothSequences([GRID] * n_local, frames=arap.INPUT_FRAMES + arap.OUTPUT_FRAMES + 2, op_frames=2,
                             seed=3 + 1000 * rank, device=device, model="dir", operators=args.operators)
    model = arap.DirModel().to(device).train()
    dp.broadcast_parameters(model, 0)
    bucket = dp.FlatGradBucket(model.parameters(), always_reduce=dist.is_initialized())
    opt = arap.make_optimizer(model)
    global_batch = n_local * world
    rng = np.random.default_rng(10 + rank)
    seq_ids = np.arange(n_local)

    def eager_step():
        batch = ds.sample_batch(n_local, rng, seq_ids=seq_ids)          # every local mesh once, random start frame
        # gradients are stored, not accumulated (.grad = None before the backward); bucket.sync() packs + all-reduces them
        return arap.train_step(model, opt, batch, global_batch=global_batch, grad_sync=bucket.sync, zero_grads=bucket.detach_grads)

    graphed = None

    def graph_step():
        batch = ds.sample_batch(n_local, rng, seq_ids=seq_ids)
        return graphed(batch, grad_sync=bucket.sync)                    # load -> one hipGraphLaunch -> pack + all-reduce -> Adam

    one_step = eager_step if args.no_graph else graph_step

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed: lazy code-object loading, allocator growth and clock ramp (3 eager steps; in graph mode the capture of
    # forward+loss+backward follows), then the W warm-up steps asked for
    for _ in range(3):
        eager_step()
    graph_fallback = None
    if not args.no_graph:
        try:
            graphed = arap.GraphedTrainStep(model, opt, ds.sample_batch(n_local, rng, seq_ids=seq_ids),
                                            global_batch=global_batch, bucket=bucket)
        except Exception as exc:  # noqa: BLE001  (a rank whose capture fails runs eagerly: same step, same collective; the line says so)
            graph_fallback = f"{type(exc).__name__}: {exc}"[:300]
            graphed = None
            args.no_graph = True
            one_step = eager_step
            try:
                torch.cuda.synchronize()
            except Exception:  # noqa: BLE001
                pass
    for _ in range(args.warmup):
        one_step()
    sync()
    timer = snF.SpmmTimer()
    ms0 = torch.cuda.memory_stats()
    t0 = time.perf_counter()
    if args.no_graph:
        n_ev = args.steps if args.spmm_timing_steps <= 0 else min(args.spmm_timing_steps, args.steps)
        with timer:                                  # per-launch HIP events on every SpMM of the first n_ev timed steps
            for i in range(n_ev):
                if i == args.linear_timing_steps:    # the ~120 Linear launches of a step carry events on the first steps only
                    timer.time_linear(False)
                loss = one_step().detach()           # (keeping the loss itself would keep the step's autograd graph alive)
        for i in range(n_ev, args.steps):            # the rest of the timed region without events
            loss = one_step().detach()
    else:
        for _ in range(args.steps):
            loss = one_step().detach()
    t_enqueue = time.perf_counter() - t0               # the host has issued every step; the device may still be running
    sync()
    dt = time.perf_counter() - t0
    ms1 = torch.cuda.memory_stats()
    alloc = {"hipMalloc_calls_in_timed_region": ms1.get("num_device_alloc", 0) - ms0.get("num_device_alloc", 0),
             "hipFree_calls_in_timed_region": ms1.get("num_device_free", 0) - ms0.get("num_device_free", 0),
             "alloc_retries": ms1.get("num_alloc_retries", 0) - ms0.get("num_alloc_retries", 0),
             "reserved_GiB": round(ms1.get("reserved_bytes.all.peak", 0) / 2**30, 2),
             "allocated_peak_GiB": round(ms1.get("allocated_bytes.all.peak", 0) / 2**30, 2)}
    if not args.no_graph:
        # kernels inside a replayed hipGraph cannot carry start/stop events, so the SpMM launches are timed on the same
        # steps launched eagerly right after the timed region (same kernels, operands and preceding kernels; the
        # rocprofv3 summary under profiles/ covers the replayed launches and agrees)
        with timer:
            for _ in range(max(1, args.roofline_steps)):
                eager_step()
        sync()
    # host cost of issuing ONE step into an idle queue (after the timed region): inside the timed loop the host runs ahead of a
    # GPU-bound step until the launch queue is full and then waits for the device — the in-loop figure measures the queue depth,
    # not the host
    host_alone = []
    for _ in range(5):
        torch.cuda.synchronize()
        th = time.perf_counter()
        one_step()
        host_alone.append(time.perf_counter() - th)
    sync()
    t_host_alone = float(np.median(host_alone))
    dt_t = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(dt_t, op=dist.ReduceOp.MAX)
    dt = float(dt_t.item())
    assert torch.isfinite(loss).item(), "training diverged"
    replicas_identical = replicas_agree(model, world)

    # ---- roofline of the dominant kernel, from the HIP events recorded during the timed steps ---------
    recs = timer.results()
    # group by KERNEL (as rocprofv3 --stats does): one template instantiation serves the four Dirac products
    # (Di, DiA forward; Di^T, DiA^T backward), which differ only in which side is the face side.
    by_kernel = {}
    for tag, M, K, nnz, N, ms in recs:
        kname = ("spmm_ring_k" if "/ring" in tag else "spmm_rb4" if "/rb4" in tag else "spmm_q3_lds" if "/q3" in tag else
                 "spmm_bsr4_lds" if "/bsr4" in tag else "spmm_csr_rows") + \
                ("_epi" if "+e" in tag else "_stats" if "+s" in tag else "") + f"<N={N}>"
        by_kernel.setdefault(kname, []).append((tag, M, K, nnz, N, ms))
    dom_name = max(by_kernel, key=lambda k: sum(r[5] for r in by_kernel[k]))
    dom = by_kernel[dom_name]
    tot_ms = sum(r[5] for r in dom)
    tot_bytes = sum(alg_bytes(r[1], r[2], r[3], r[4], r[0]) for r in dom)
    avg_ms = tot_ms / len(dom)
    ab = tot_bytes / len(dom)                      # average algorithmic bytes per launch of this kernel
    achieved = tot_bytes / (tot_ms * 1e-3)
    n_ev_steps = (args.steps if args.spmm_timing_steps <= 0 else min(args.spmm_timing_steps, args.steps)) if args.no_graph else max(1, args.roofline_steps)
    spmm_ms_per_step = sum(r[5] for r in recs) / n_ev_steps
    shapes = {}
    for tag, M, K, nnz, N, ms in dom:
        shapes.setdefault((tag, M, K, nnz, N), []).append(ms)
    per_shape = [{"product": t, "M": M, "K": K, "nnz": nnz, "N": N, "launches": len(v), "avg_launch_ms": float(np.mean(v)),
                  "algorithmic_bytes": alg_bytes(M, K, nnz, N, t), "frac": alg_bytes(M, K, nnz, N, t) / (float(np.mean(v)) * 1e-3) / HBM_PEAK}
                 for (t, M, K, nnz, N), v in shapes.items()]
    tag = dom[0][0]

    # HBM traffic of that kernel: rocprofv3 cannot run inside this process, so the figure is the committed PMC measurement
    # of THIS command (profiles/r3_pmc_traffic_c3.json: rocprofv3 --pmc over bench.py itself, i.e. in-step launches with
    # the step's own predecessors and cache state; raw per-dispatch counters next to it), averaged over the kernel's launches.
    # The file carries the sha256 of the kernel sources it was measured on (tools/pmc_bench.py): on any other sources the
    # figure is withheld (traffic: null) instead of quoted stale.
    traffic = traffic_src = None
    step_traffic = None
    pmc_file = os.path.join("profiles", "r6_pmc_traffic_c3.json")
    table, in_run_note = (None, "not requested")
    if rank == 0 and world == 1 and not args.no_pmc:
        # release the device memory of the timed run first: the counter passes are child processes on the same GPU
        torch.cuda.empty_cache()
        table, in_run_note = pmc_in_run(args)
    try:
        if table is None:
            with open(os.path.join(ROOT, pmc_file)) as fh:
                table = json.load(fh)
            file_note = f"{pmc_file} (kernel sources {table.get('_csrc_sha256', '?')[:12]}; in-run pass: {in_run_note})"
        else:
            file_note = None
            table["_csrc_sha256"] = csrc_digest()
        if table.get("_csrc_sha256") != csrc_digest():
            traffic_src = f"{pmc_file} was measured on other kernel sources (sha256 mismatch): withheld; in-run pass: {in_run_note}"
        elif file_note is not None and (args.meshes != MESHES_PER_GPU or args.format != "q3"):
            # the committed file is the default configuration's (64 meshes per GPU, quaternion-packed operators): any other batch
            # or storage form has other launches, so nothing is quoted rather than that file's bytes over this run's durations
            traffic_src = (f"{pmc_file} holds the default configuration ({MESHES_PER_GPU} meshes per GPU, q3), this run is "
                           f"{args.meshes} meshes / {args.format}: withheld; in-run pass: {in_run_note}")
        else:
            tot = table.get("_total_bytes_counted")
            n_counted = table.get("_steps_counted", 4)
            if tot:
                step_traffic = {"GB_per_step": (tot["read"] + tot["write"]) / n_counted / 1e9, "read_GB": tot["read"] / n_counted / 1e9,
                                "write_GB": tot["write"] / n_counted / 1e9, "steps_counted": n_counted}
            base = dom_name.split("<")[0]
            # (the library launches every Q3 kernel in two shapes, <name> and <name>_wide; the timing tags do not distinguish them)
            hits = [v for k, v in table.items() if not k.startswith("_") and k.split("<")[0] in (base, base + "_wide") and f"<{dom[0][4]}," in k]
            if hits:
                n_l = sum(h["launches"] for h in hits)
                traffic = sum((h["read_bytes_mean"] + h["write_bytes_mean"]) * h["launches"] for h in hits) / n_l
                traffic_src = ((file_note + ": rocprofv3 --pmc TCC_EA0_RDREQ_sum / TCC_EA0_WRREQ_sum over this bench command (in-step), "
                                "bytes = RDREQ*128 + WRREQ*64") if file_note else in_run_note) + f"; mean over {n_l} launches of {base}"
    except (OSError, ValueError, KeyError):
        if traffic_src is None:
            traffic_src = f"no counter data: in-run pass: {in_run_note}; {pmc_file} not readable"
    product_bytes = sum(alg_bytes(r[1], r[2], r[3], r[4], "") for r in dom)       # SURVEY §8(d) bytes of the products alone
    frac_alg = achieved / HBM_PEAK
    frac_meas = (traffic / (avg_ms * 1e-3) / HBM_PEAK) if traffic else None
    if frac_meas is not None and frac_meas < frac_alg:
        frac_top, frac_conv = frac_meas, "measured_hbm_traffic (PMC, in-step) / average launch duration / peak"
    else:
        frac_top, frac_conv = frac_alg, "product_plus_epilogue_operands (algorithmic bytes) / average launch duration / peak"
    # the Linear-layer kernels of the same steps (forward / input gradient / weight gradient launchers record themselves)
    lin = {}
    for name, rows_, width, outw, nbytes, ms_ in getattr(timer, "linear", []):
        lin.setdefault((name, rows_, width, outw, nbytes), []).append(ms_)
    n_steps_timed = n_ev_steps
    n_lin_steps = min(n_steps_timed, max(1, args.linear_timing_steps)) if args.no_graph else n_steps_timed
    linear_kernels = sorted(({"kernel": k[0], "rows": k[1], "width": k[2], "out_width": k[3], "launches": len(v),
                              "launches_per_step": len(v) / n_lin_steps, "avg_ms": float(np.mean(v)), "bytes": k[4],
                              "TBps": k[4] / (float(np.mean(v)) * 1e-3) / 1e12, "frac": k[4] / (float(np.mean(v)) * 1e-3) / HBM_PEAK,
                              "ms_per_step": float(np.sum(v)) / n_lin_steps} for k, v in lin.items()),
                            key=lambda d: -d["ms_per_step"])

    out = {
        "metric": "meshes/sec fwd+bwd, Dirac temporal-predict",
        "value": global_batch * args.steps / dt,
        "unit": "meshes/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"as_rigid_as_possible Dirac temporal prediction: {n_local} grid-cloth meshes {GRID[0]}x{GRID[1]} "
                               f"(V=5041,F=9800) per GPU, C=128, 15 layers, fwd+loss+bwd+allreduce+Adam",
                   "meshes_per_gpu": n_local, "global_batch": global_batch, "parallelism": f"dp{world} (mesh sharding, flat-bucket RCCL all-reduce)",
                   "world_size": world,
                   "rccl_ranks": world if (dist.is_initialized() and dist.get_backend() == "nccl") else 0,
                   "collective_backend": (dist.get_backend() if dist.is_initialized() else (group_note or "none (single rank, no process group)")),
                   "collective_per_step": ((f"pack + ncclAllReduce(SUM) of the {bucket.nbytes}-byte flat gradient bucket over {world} rank(s)"
                                            if dist.get_backend() == "nccl" else f"pack + {dist.get_backend()} all-reduce of the flat bucket")
                                           if dist.is_initialized() else "none"),
                   "devices_visible": torch.cuda.device_count(), "ranks_share_devices": bool(oversubscribed),
                   "replicas_identical_after_the_run": replicas_identical,
                   "linear_layers": ("fp32 operands and fp32 accumulation; products formed on the 16-bit matrix pipe from an exact split of "
                                     "every operand into two fp16 pieces after a power-of-two row / column scaling (3 partial products, "
                                     "error <= 2^-23 per term: fp32-accurate, tests/test_dense_gpu.py); SN_GEMM_VARIANT=1 selects the "
                                     "three-piece bf16 form, 0 the fp32-MFMA kernels; the weight gradient (wgrad_h_k) uses two fp16 pieces of each "
                                     "operand with the row / column bounds the step already holds, the first layer's (K = 6) wgrad_u_k"),
                   "allocator": alloc, "operator_format": args.format, "operators": args.operators,
                   "host_enqueue_ms_per_step": t_host_alone * 1e3,
                   "host_enqueue_definition": "host time to issue one step into an idle launch queue (median of 5, after the timed "
                                              "region); in_loop: the same inside the timed loop, where a GPU-bound step makes the host "
                                              "wait for queue space",
                   "host_enqueue_in_loop_ms_per_step": t_enqueue / args.steps * 1e3, "host_affinity": affinity,
                   "launch": ("eager: every block direction one launch plan (sn_plan_run), the rest launched from Python"
                              if plans.enabled() else "eager: every kernel launched from Python (SN_PLANS=0)") if args.no_graph
                   else "hipGraph replay of fwd+loss+bwd; sampling, all-reduce, Adam eager",
                   "launch_plans": {k: v["replayed"] for k, v in plans.stats().items() if v["replayed"] or v["refused"]},
                   "plan_graph_launches": plans.graph_stats()["launched"],
                   "graph_fallback": graph_fallback, "grad_bucket_bytes": bucket.nbytes},
        "roofline": {"bound": "hbm", "kernel": dom_name + (" (the backward products Di^T, DiA^T, ELU backward fused into the store)" if "_epi" in dom_name
                                                          else " (the Dirac products launched without epilogue)"),
                     # `frac` is the MORE CONSERVATIVE of the two defensible readings of the fused launch: algorithmic bytes of the
                     # product plus the epilogue operands it must read, or — when a counter file of these very sources exists —
                     # the HBM traffic the counters measured; the three conventions stay side by side below
                     "achieved": frac_top * HBM_PEAK / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": frac_top,
                     "frac_convention": frac_conv,
                     "traffic": traffic, "traffic_source": traffic_src, "step_traffic": step_traffic,
                     "algorithmic_bytes_definition": "SURVEY.md §8(d) CSR/int32/fp32 bytes of the product (nnz*8 + (M+1)*4 + K*N*4 + M*N*4) "
                                                     "plus, for the fused ELU-backward launches, the epilogue operands E and G (M*N*4 each) "
                                                     "that the fused kernel must read; the three figures below separate the conventions",
                     "frac_by_convention": {
                         "product_plus_epilogue_operands": frac_alg,
                         "product_bytes_only": product_bytes / (tot_ms * 1e-3) / HBM_PEAK,
                         "measured_hbm_traffic": frac_meas},
                     "algorithmic_bytes_per_launch": ab, "avg_launch_ms": avg_ms,
                     "timing": "hipExtLaunchKernelGGL start/stop events on the launch stream, " +
                               (f"every launch of the first {n_ev_steps} of the {args.steps} timed steps" if args.no_graph else
                                f"every launch of {max(1, args.roofline_steps)} eager steps run right after the timed hipGraph replays"),
                     "launches_timed": len(dom), "spmm_ms_per_step_all_kernels": spmm_ms_per_step, "per_product": per_shape,
                     "other_spmm_kernels": [
                         {"kernel": k, "launches": len(v), "avg_launch_ms": sum(r[5] for r in v) / len(v),
                          "frac": sum(alg_bytes(r[1], r[2], r[3], r[4], r[0]) for r in v) / (sum(r[5] for r in v) * 1e-3) / HBM_PEAK}
                         for k, v in by_kernel.items() if k != dom_name],
                     "linear_kernels": linear_kernels,
                     "linear_kernels_note": "kernel[variant]: linear_fwd bits 1 elu copy, 2 residual, 4 y written, 8 per-mesh bias; "
                                            "linear_dgrad bits 1 BatchNorm tail, 2 through the activation, 4 gadd, 8 per-mesh vector; "
                                            "bytes = operands read + results written (weights excluded), timed like the SpMM launches"
                                            + (f" on the first {n_lin_steps} of the {args.steps} timed steps (an event pair costs a launch ~2 us; "
                                               f"the sparse products on the first {n_ev_steps})" if args.no_graph else ""),
                     "linear_ms_per_step": float(sum(d["ms_per_step"] for d in linear_kernels))},
    }
    if not args.no_secondary:
        # config 5: every rank runs its own replica of the microbench (no collective on this path: aggregate = sum)
        del ds, model, opt, bucket, graphed
        torch.cuda.empty_cache()
        try:
            sec = c5_secondary(device, rank)
            mine = sec["GBps_mean_packed"]
        except Exception as exc:  # noqa: BLE001 — the headline line must survive a failure of the secondary block
            sec, mine = {"error": repr(exc)[:300]}, 0.0
        agg = torch.tensor([mine], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(agg)                  # (every rank reaches this, failed or not)
        sec["aggregate_GBps_all_ranks_packed_mean"] = float(agg.item())
        # BASELINE configs[3] as the sharded job it is: one pair per rank and step, every rank takes part (also at N = 1)
        try:
            del ds, model, opt, bucket
        except NameError:
            pass
        torch.cuda.empty_cache()
        try:
            dt_f, info = c4_dp_step(device, rank, world, 30, 6)
            t = torch.tensor([dt_f], dtype=torch.float64, device=device)
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_f = float(t.item())
            sec["config4_dp"] = {"workload": "BASELINE configs[3]: FAUST dense correspondence, one pair (2 x 6890 vertices padded to 7000) per "
                                             "rank and step, hipGraph replay + flat-bucket all-reduce + Adam; max over ranks",
                                 "n_gpus": world, "ms_per_step": dt_f * 1e3, "pairs_per_s": world / dt_f, "meshes_per_s": 2 * world / dt_f, **info}
        except Exception as exc:  # noqa: BLE001
            sec["config4_dp"] = {"error": repr(exc)[:300]}
            if world > 1:
                raise                               # (a rank that left the collectives would hang the others)
        if rank == 0 and world == 1 and not args.no_pmc and "products" in sec:
            torch.cuda.empty_cache()
            tr5, note5 = c5_pmc_traffic()
            sec["traffic_source"] = note5
            if tr5:
                for p_ in sec["products"] + sec["laplacian"]:
                    t_ = tr5.get((p_["order"], p_["product"])) if p_["layout"] == "packed" else None
                    p_["traffic"] = t_
                    p_["frac_traffic"] = (t_ / (p_["ms_median"] * 1e-3) / HBM_PEAK) if t_ else None
        if rank == 0 and world == 1:
            # the small-batch configurations, driver-visible (rank 0 of a one-GPU run only: they are replicas, not a sharded job)
            for key, fn in (("config3_order", c3_order_secondary), ("config3_swap", c3_swap_secondary), ("config2", c2_secondary),
                            ("config4_pair", c4_pair_secondary), ("config2_swap", c2_swap_secondary), ("config4_swap", c4_swap_secondary)):
                torch.cuda.empty_cache()
                try:
                    sec[key] = fn(device)
                except Exception as exc:  # noqa: BLE001
                    sec[key] = {"error": repr(exc)[:300]}
        out["secondary"] = sec
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(seed=3)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        emit(out, result_fd)
    if dist.is_initialized():
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
