"""Step-time measurements of the other BASELINE configs (parity-test cases, not bench lines):
   python tools/train_bench.py mnist_dir|mnist_lap|faust_lap|arap_lap|arap_ragged|arap_swap|mnist_swap|faust_swap [steps]
mnist_dir = config 2 (Mesh-MNIST Dirac, batch 512); mnist_lap = config 1 shape on the GPU; faust_lap = config 4 per-GPU
work (one pair of 6890-vertex bodies padded to 7000); arap_lap = the Laplacian variant of config 3; arap_swap = config 3
as an UNMODIFIED reference driver runs it after the import swap: per step and per sample sp_sparse_to_pt_sparse,
sparse_diag_cat, .cuda() of operators / inputs / targets, the reference's model calling sequence and loss
(src/as_rigid_as_possible/main.py:156-232) — served from the resident cache behind those names (surfacenetworks_amd/resident.py);
SN_RESIDENT=0 runs the reference's host arithmetic instead (use SN_SWAP_MESHES=8: 3 s per step at 64);
SN_SWAP_MODEL=product puts the product's own DirModel on the same batches."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surfacenetworks_amd import dp, arap, dense_correspondence as dc, mesh_mnist as mm  # noqa: E402


def timed(step, steps, warm=3):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def arap_swap(dev="cuda", steps=10, B=64, permute=False):
    """BASELINE configs[2] through the reference's own batching names (module docstring: arap_swap)."""
    import torch.nn as nn
    import torch.nn.functional as F

    import surfacenetworks_amd.utils_pt as utils
    from surfacenetworks_amd import mesh_ops
    from surfacenetworks_amd.resident import resident_cache

    class RefStyleDirModel(nn.Module):
        """The calling sequence of the reference's DirModel (src/as_rigid_as_possible/models.py:108-152) on the swapped
        `utils`: zero face features as a tensor, every block with the reference's four arguments, conv2(F.elu(v)), the
        repeated last frame — none of the product's own hints (f=None, f_out_needed, avg_next, elu_conv1x1)."""

        def __init__(self):
            super().__init__()
            self.conv1 = utils.GraphConv1x1(6, 128, batch_norm=None)
            for i in range(15):
                self.add_module("rn{}".format(i), utils.DirResNet2(128) if i % 2 == 0 else utils.AvgResNet2(128))
            self.do = nn.Dropout2d()
            self.conv2 = utils.GraphConv1x1(128, 120, batch_norm="pre")

        def forward(self, Di, DiA, mask, inputs):
            batch_size, num_nodes, _ = inputs.size()
            v = self.conv1(inputs)
            num_faces = DiA.size(2) // 4 if len(Di.size()) == 3 else DiA.size(1) // 4 // batch_size
            f = torch.zeros(batch_size, num_faces, 128, device=v.device)
            for i in range(15):
                if i % 2 == 0:
                    v, f = self._modules["rn{}".format(i)](Di, DiA, v, f)
                else:
                    v = self._modules["rn{}".format(i)](None, mask, v)
            x = self.conv2(F.elu(v))
            return x + inputs[:, :, -3:].repeat(1, 1, 40)

    grid_rng = np.random.default_rng(3)
    samples = []
    for _ in range(B):                                          # the dataset as the reference keeps it: scipy operators per frame
        V, F_ = mesh_ops.grid_cloth(71, 71, grid_rng, permute="both" if permute else False)
        Di, DiA = mesh_ops.dirac(V, F_)
        samples.append((V.astype(np.float32), Di.astype(np.float32), DiA.astype(np.float32), F_.shape[0]))
    nv, nf = samples[0][0].shape[0], samples[0][3]
    model = (arap.DirModel() if os.environ.get("SN_SWAP_MODEL", "ref") == "product" else RefStyleDirModel()).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), 1e-3, weight_decay=1e-5)                    # main.py:207
    inputs = torch.from_numpy(np.stack([np.concatenate([s_[0], s_[0]], 1) for s_ in samples]))
    targets = torch.zeros(B, nv, 120)
    mask = torch.ones(B, nv, 1)
    t_host, t_h2d = [], []

    def step():
        t0 = time.perf_counter()
        Di = utils.sparse_diag_cat([utils.sp_sparse_to_pt_sparse(s_[1]) for s_ in samples], 4 * nf, 4 * nv)     # main.py:161-181
        DiA = utils.sparse_diag_cat([utils.sp_sparse_to_pt_sparse(s_[2]) for s_ in samples], 4 * nv, 4 * nf)
        t1 = time.perf_counter()
        Di, DiA, x, y, m = Di.cuda(), DiA.cuda(), inputs.cuda(), targets.cuda(), mask.cuda()                 # main.py:184
        t2 = time.perf_counter()
        t_host.append(t1 - t0), t_h2d.append(t2 - t1)
        outputs = model(Di, DiA, m, x)                                                                         # main.py:222-232
        outputs = outputs * m.expand_as(outputs)
        loss = F.smooth_l1_loss(outputs, y, reduction="sum") / B
        opt.zero_grad()
        loss.backward()
        opt.step()
    dt = timed(step, steps, warm=2)
    k = len(t_host) - steps
    c = resident_cache()
    return {"workload": f"config 3 as an UNMODIFIED reference driver runs it after the import swap: per step {2 * B} x sp_sparse_to_pt_sparse, "
                        "2 x sparse_diag_cat, .cuda() of operators / inputs / targets / mask (host tensors, pageable), the reference's "
                        "DirModel calling sequence, mask multiply + smooth_l1_loss + Adam (src/as_rigid_as_possible/main.py:156-232)",
            "meshes": B, "model": type(model).__name__, "vertex_order": "shuffled (vertices and faces)" if permute else "grid", "resident": os.environ.get("SN_RESIDENT", "1") != "0",
            "steps": steps, "ms_per_step": dt * 1e3, "meshes_per_s": B / dt,
            "host_batching_calls_ms": float(np.mean(t_host[k:]) * 1e3), "driver_cuda_calls_host_ms": float(np.mean(t_h2d[k:]) * 1e3),
            "pageable_MB_per_step": (inputs.numel() + targets.numel() + mask.numel()) * 4 / 1e6,
            "resident_cache": None if c is None else {"hits": c.hits, "misses": c.misses,
                                                      "MB_in_HBM": sum(p_.device_bytes() for p_ in c.pools.values()) / 1e6}}


def mnist_swap(dev="cuda", steps=20, B=512, permute=False):
    """BASELINE configs[1] as the reference's Mesh-MNIST loop runs it after the import swap (src/mesh_mnist/main.py:79-117,151-167):
    per step utils.sparse_cat of the batch's per-sample Di / DiA handles (3-D operators), .cuda() of inputs / targets / mask /
    operators, DirModel(inputs, Di, DiA, mask), F.nll_loss, Adam — eager, nothing captured.
    permute: the meshes' vertices and faces in random order, used as stored (the Delaunay generator's order otherwise)."""
    import torch.nn.functional as F

    import surfacenetworks_amd.utils_pt as utils
    from surfacenetworks_amd import mesh_ops

    rng = np.random.default_rng(2)
    samples = []
    for _ in range(B):
        V, F_ = mesh_ops.delaunay_disc(150, rng)
        if permute:
            pv, pf = rng.permutation(V.shape[0]), rng.permutation(F_.shape[0])
            inv = np.empty_like(pv)
            inv[pv] = np.arange(pv.size)
            V, F_ = V[pv], inv[F_][pf]
        Di, DiA = mesh_ops.dirac(V, F_)
        samples.append({"V": torch.from_numpy(V.astype(np.float32)), "F": F_, "label": int(rng.integers(0, 10)),
                        "Di": utils.sp_sparse_to_pt_sparse(Di.astype(np.float32)), "DiA": utils.sp_sparse_to_pt_sparse(DiA.astype(np.float32))})
    nv = max(s_["V"].shape[0] for s_ in samples)
    nf = max(s_["F"].shape[0] for s_ in samples)
    model = mm.DirModel().to(dev).train()
    opt = torch.optim.Adam(model.parameters(), 1e-3, weight_decay=1e-5)                    # main.py:139
    inputs, targets, mask = torch.zeros(B, nv, 3), torch.zeros(B).long(), torch.zeros(B, nv, 1)
    for b, s_ in enumerate(samples):
        inputs[b, : s_["V"].shape[0]] = s_["V"]
        targets[b] = s_["label"]
        mask[b, : s_["V"].shape[0]] = 1

    def step():
        Di = utils.sparse_cat([s_["Di"] for s_ in samples], 4 * nf, 4 * nv)                                   # main.py:109-111
        DiA = utils.sparse_cat([s_["DiA"] for s_ in samples], 4 * nv, 4 * nf)
        x, y, m, Di, DiA = inputs.cuda(), targets.cuda(), mask.cuda(), Di.cuda(), DiA.cuda()                   # main.py:115
        loss = F.nll_loss(model(x, Di, DiA, m), y)                                                             # main.py:157-160
        opt.zero_grad()
        loss.backward()
        opt.step()
    dt = timed(step, steps, warm=14)
    return {"workload": "config 2 as the reference's Mesh-MNIST loop runs it after the import swap: per step 2 x utils.sparse_cat of 512 "
                        "per-sample handles, .cuda() of inputs / targets / mask / operators, DirModel, nll_loss, Adam (src/mesh_mnist/"
                        "main.py:79-117,151-167); eager", "vertex_order": "shuffled" if permute else "as generated",
            "steps": steps, "ms_per_step": dt * 1e3, "meshes_per_s": B / dt}


def faust_swap(dev="cuda", steps=20, permute=False):
    """BASELINE configs[3] (per-GPU work: one pair) as the reference's dense-correspondence loop runs it after the import swap
    (src/dense_correspondence/main.py:106-191,310-327): per sample utils.sparse_diag_cat([L], 7000, 7000).coalesce().cuda(),
    zero-padded inputs / mask .cuda(), SiameseModel(lap) -> (1, 7000, 7000) scores, loss_fun_delta_cross_entropy, Adam — eager.
    The geodesic matrices and label vectors stay on the device (the reference moves them with .cuda() per step from wherever
    its loader left them).  permute: vertices and faces of the bodies in random order, used as stored."""
    import surfacenetworks_amd.utils_pt as utils
    from surfacenetworks_amd import mesh_ops

    rng = np.random.default_rng(4)
    frames = []
    for _ in range(4):
        V, F_ = mesh_ops.torus_grid(65, 106, rng)
        if permute:
            pv, pf = rng.permutation(V.shape[0]), rng.permutation(F_.shape[0])
            inv = np.empty_like(pv)
            inv[pv] = np.arange(pv.size)
            V, F_ = V[pv], inv[F_][pf]
        nv = V.shape[0]
        label = rng.permutation(nv)
        Vd = torch.from_numpy(V.astype(np.float32)).to(dev)
        frames.append({"V": torch.from_numpy(V.astype(np.float32)), "L": utils.sp_sparse_to_pt_sparse(mesh_ops.laplacian(V, F_).astype(np.float32)),
                       "G": torch.cdist(Vd, Vd), "label": torch.from_numpy(label).to(dev), "label_inv": torch.from_numpy(np.argsort(label)).to(dev)})
    model = dc.SiameseModel("lap", 15).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), 1e-3, weight_decay=1e-5)                    # main.py:285
    k = [0]

    def sample(i):
        fr = frames[i]
        nv = fr["V"].shape[0]
        inputs, mask = torch.zeros(1, 7000, 3), torch.zeros(1, 7000, 1)
        inputs[0, :nv] = fr["V"]
        mask[0, :nv] = 1
        L = utils.sparse_diag_cat([fr["L"]], 7000, 7000).coalesce()                                           # main.py:180
        return inputs.cuda(), [(fr["G"].cuda(), fr["label"].cuda(), fr["label_inv"].cuda())], mask.cuda(), L.cuda()

    def step():
        k[0] += 1
        inX, tX, mX, LX = sample(k[0] % 4)
        inY, tY, mY, LY = sample((k[0] + 1) % 4)
        out = model((LX, mX), (LY, mY), inX, inY)                                                              # main.py:317-321
        loss = dc.loss_fun_delta_cross_entropy(out, tX, tY)
        opt.zero_grad()
        loss.backward()
        opt.step()
    dt = timed(step, steps, warm=52)
    return {"workload": "config 4 (one pair) as the reference's dense-correspondence loop runs it after the import swap: per sample "
                        "utils.sparse_diag_cat([L], 7000, 7000).coalesce().cuda(), padded inputs / mask .cuda(), SiameseModel(lap), "
                        "loss_fun_delta_cross_entropy on the (1, 7000, 7000) scores, Adam (src/dense_correspondence/main.py:106-191,"
                        "310-327); eager", "vertex_order": "shuffled" if permute else "as generated",
            "steps": steps, "ms_per_step": dt * 1e3, "pairs_per_s": 1 / dt, "meshes_per_s": 2 / dt}


def main():
    what = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = "cuda"
    rng = np.random.default_rng(0)
    torch.manual_seed(0)
    if what in ("mnist_dir", "mnist_lap"):
        B = 512 if what == "mnist_dir" else 32
        ds = mm.MeshDigits(B, seed=2, device=dev, fixed_vertices=150 if what == "mnist_dir" else None,
                           model="dir" if what == "mnist_dir" else "lap")
        model = (mm.DirModel() if what == "mnist_dir" else mm.Model()).to(dev).train()
        opt = mm.make_optimizer(model)
        ids = np.arange(B)
        dt = timed(lambda: mm.train_step(model, opt, ds.sample_batch(B, rng, ids=ids)), steps, warm=14)
        ex = ds.sample_batch(B, rng, ids=ids)
        try:
            g = mm.graphed_train_step(model, opt, ex)
            nb = ds.sample_batch(B, rng, ids=ids)
            if g.matches(nb):
                dtg = timed(lambda: g(ds.sample_batch(B, rng, ids=ids)), steps)
                print(f"{what}: hipGraph replay of fwd+loss+bwd: {dtg * 1e3:.2f} ms/step, {B / dtg:.0f} meshes/s")
                from surfacenetworks_amd.graphs import BatchAhead

                ahead = BatchAhead(lambda: ds.sample_batch(B, rng, ids=ids), dev)
                dta = timed(lambda: g(ahead.get()), steps)
                print(f"{what}: the same with the batch assembled one step ahead on a side stream: {dta * 1e3:.2f} ms/step, {B / dta:.0f} meshes/s")
            else:
                print(f"{what}: batches have varying signatures (ragged meshes): no graph replay")
        except Exception as exc:  # noqa: BLE001
            print(f"{what}: graph capture failed: {exc}")
        print(f"{what}: batch {B}, {dt * 1e3:.2f} ms/step, {B / dt:.0f} meshes/s")
    elif what == "faust_lap":
        ds = dc.TorusBodies(4, device=dev)
        model = dc.SiameseModel("lap", 15).to(dev).train()
        opt = dc.make_optimizer(model)
        k = [0]

        def step():
            k[0] += 1
            dc.train_step(model, opt, ds, k[0] % 4, (k[0] + 1) % 4)
        dt = timed(step, steps, warm=52)
        print(f"{what}: 1 pair of 6890-vertex bodies (padded 7000), {dt * 1e3:.2f} ms/step, {2 / dt:.1f} meshes/s")
        try:
            g = dc.graphed_train_step(model, opt, dc.PairBatch(ds, 0, 1))

            def gstep():
                k[0] += 1
                g(dc.PairBatch(ds, k[0] % 4, (k[0] + 1) % 4))
            dtg = timed(gstep, steps)
            print(f"{what}: hipGraph replay of fwd+loss+bwd: {dtg * 1e3:.2f} ms/step, {2 / dtg:.1f} meshes/s")
        except Exception as exc:  # noqa: BLE001
            print(f"{what}: graph capture failed: {exc!r}")
    elif what == "arap_lap":
        ds = arap.ClothSequences([(71, 71)] * 64, frames=44, op_frames=2, seed=3, device=dev, model="lap")
        model = arap.Model(15).to(dev).train()
        opt = arap.make_optimizer(model)
        ids = np.arange(64)
        bucket = dp.FlatGradBucket(model.parameters())          # stored gradients, as bench.py's step
        dt = timed(lambda: arap.train_step(model, opt, ds.sample_batch(64, rng, seq_ids=ids), grad_sync=bucket.sync,
                                           zero_grads=bucket.detach_grads), steps)
        print(f"{what}: batch 64 x 71x71, {dt * 1e3:.2f} ms/step, {64 / dt:.0f} meshes/s")
    elif what == "arap_ragged":
        # a ragged batch (config-5-like sizes: 64 cloth meshes with 1 000 .. 10 000 vertices): padded as the reference batches
        # (every mesh to the batch maximum) against PACKED (no padding rows, PackedSegments; BatchNorm over real rows only)
        vs = np.random.default_rng(5).integers(1000, 10001, size=64)
        grids = [(int(np.sqrt(v)), int(v) // int(np.sqrt(v))) for v in vs]
        ds = arap.ClothSequences(grids, frames=44, op_frames=2, seed=3, device=dev, model="dir")
        ids = np.arange(64)
        for packed in (False, True):
            torch.manual_seed(0)
            model = arap.DirModel().to(dev).train()
            opt = arap.make_optimizer(model)
            bucket = dp.FlatGradBucket(model.parameters())
            dt = timed(lambda: arap.train_step(model, opt, ds.sample_batch(64, rng, seq_ids=ids, packed=packed),
                                               grad_sync=bucket.sync, zero_grads=bucket.detach_grads), steps)
            rows = int(ds.num_vertices.sum()) if packed else 64 * int(ds.num_vertices.max())
            print(f"{what}: 64 ragged meshes (sum V = {int(ds.num_vertices.sum())}, max V = {int(ds.num_vertices.max())}), "
                  f"{'packed' if packed else 'padded'}: {rows} vertex rows, {dt * 1e3:.2f} ms/step, {64 / dt:.0f} meshes/s")
    elif what == "arap_swap":
        r = arap_swap(dev, steps, int(os.environ.get("SN_SWAP_MESHES", "64")))
        print(f"{what}: batch {r['meshes']} x 71x71 behind the reference's names (resident={r['resident']}, model {r['model']}): "
              f"{r['ms_per_step']:.1f} ms/step, {r['meshes_per_s']:.1f} meshes/s (host batching calls {r['host_batching_calls_ms']:.2f} ms, the driver's "
              f".cuda() calls incl. {r['pageable_MB_per_step']:.0f} MB of pageable inputs/targets {r['driver_cuda_calls_host_ms']:.2f} ms host time; "
              f"resident cache {r['resident_cache']})")
    elif what in ("mnist_swap", "faust_swap"):
        fn = mnist_swap if what == "mnist_swap" else faust_swap
        r = fn(dev, steps)
        r2 = fn(dev, steps, permute=True)
        print(f"{what}: {r['ms_per_step']:.2f} ms/step ({r['meshes_per_s']:.0f} meshes/s) behind the reference's names, eager; vertices and "
              f"faces shuffled, as stored: {r2['ms_per_step']:.2f} ms/step")
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
