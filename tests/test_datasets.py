"""Readers of the reference's on-disk formats: written in the reference layout by the synthetic writers, read back into
resident pools (kernels replaced by the CPU seam), and checked against direct construction."""
import numpy as np
import torch

from surfacenetworks_amd import arap, datasets, mesh_ops


def test_arap_sequence_roundtrip(tmp_path, cpu_kernels):
    rng = np.random.default_rng(0)
    paths = []
    for s, (n, m) in enumerate([(6, 5), (7, 6)]):
        V0, F = mesh_ops.grid_cloth(n, m, rng)
        Vt = np.stack([V0 + 0.01 * t * np.array([0, 0, 1.0]) * np.sin(V0[:, :1] * 6) for t in range(44)])
        p = str(tmp_path / f"seq{s}.npy")
        datasets.write_arap_sequence(p, Vt, F, op_frames=3)
        paths.append(p)
    seq = datasets.load_arap_sequence(paths[0])
    assert len(seq) == 44 and seq[0]["Di"].shape == (4 * seq[0]["F"].shape[0], 4 * seq[0]["V"].shape[0]) and "Di" not in seq[5]
    ds = datasets.arap_from_files(paths, device="cpu", model="dir")
    assert ds.op_frames == 3 and ds.frames == 44 and ds.n == 2
    b = ds.sample_batch(2, None, seq_ids=np.array([1, 0]), offsets=np.array([1, 0]))
    nv, nf = 42, int(ds.num_faces.max())
    assert b.inputs.shape == (2, nv, 6) and b.Di.shape == (2 * 4 * nf, 2 * 4 * nv)
    # operator of sample 0 = sequence 1, frame offset + 1 = 2
    want = datasets.load_arap_sequence(paths[1])[2]["Di"]
    got = b.Di.to_scipy()[: want.shape[0], : want.shape[1]]
    assert abs(got - want).max() == 0
    m = arap.DirModel()
    loss, out = arap.forward_loss(m, b)
    assert torch.isfinite(loss)


def test_mesh_mnist_roundtrip(tmp_path, cpu_kernels):
    from surfacenetworks_amd import mesh_mnist

    rng = np.random.default_rng(1)
    meshes = [mesh_ops.delaunay_disc(int(n), rng) for n in (30, 41, 36)]
    p = str(tmp_path / "train_plus.np")
    datasets.write_mesh_mnist(p, meshes, [3, 1, 4])
    samples = datasets.load_mesh_mnist(p)
    assert set(samples[0]) == {"V", "F", "L", "flat_L", "Di", "DiA", "flat_Di", "flat_DiA", "label"}
    ds = datasets.mnist_from_samples(samples, device="cpu", model="lap")
    b = ds.sample_batch(3, None, ids=np.array([2, 0, 1]))
    assert b.targets.tolist() == [4, 3, 1] and b.inputs.shape == (3, 41, 3)
    # stored in the locality numbering (Delaunay vertices come in random order): the file's operator, renumbered
    o = ds.orders[2]
    assert not o.identity
    want = mesh_ops.permute_operator(samples[2]["L"], o.vorder, o.vorder)
    assert abs(b.L.to_scipy()[: want.shape[0], : want.shape[1]] - want).max() == 0
    assert np.array_equal(b.inputs[0, : o.vorder.size].numpy(), samples[2]["V"][o.vorder])
    raw = datasets.mnist_from_samples(samples, device="cpu", model="lap", reorder=False)
    b0 = raw.sample_batch(3, None, ids=np.array([2, 0, 1]))
    assert abs(b0.L.to_scipy()[: want.shape[0], : want.shape[1]] - samples[2]["L"]).max() == 0
    loss, out = mesh_mnist.forward_loss(mesh_mnist.Model(), b)
    assert out.shape == (3, 10) and torch.isfinite(loss)


def test_faust_frame_roundtrip(tmp_path):
    rng = np.random.default_rng(2)
    V, F = mesh_ops.torus_grid(6, 8, rng)
    label = rng.permutation(V.shape[0])
    G = np.abs(rng.standard_normal((V.shape[0], V.shape[0])))
    p = str(tmp_path / "tr_reg_000.npz")
    datasets.write_faust_frame(p, V, F, label, G)
    fr = datasets.load_faust_frame(p, device="cpu")
    assert fr["L"].shape == (48, 48) and fr["Di"].shape == (4 * 96, 4 * 48)
    assert torch.equal(fr["label_inv"][fr["label"]], torch.arange(48))
    assert abs(fr["L"] - mesh_ops.laplacian(V, F).astype("f")).max() == 0


def test_reference_format_files_feed_training(golden_dir, cpu_kernels):
    """Fixture files written by the reference's own add_laplacian.process (tests/golden/make_dataset_fixtures.py)."""
    import product_checks as pc

    pc.check_dataset_files(golden_dir, "cpu")


def test_streamed_faust_loss_equals_materialised(cpu_kernels):
    """Streamed CE over row blocks == loss_fun_delta_cross_entropy on bmm(FA, FB^T): value and both gradients."""
    from surfacenetworks_amd import dense_correspondence as dc

    torch.manual_seed(0)
    NA = NB = 333
    FA = torch.randn(1, NA, 120, dtype=torch.float64, requires_grad=True)
    FB = torch.randn(1, NB, 120, dtype=torch.float64, requires_grad=True)
    GA, GB = torch.rand(NA, NA, dtype=torch.float64), torch.rand(NB, NB, dtype=torch.float64)
    lA, lB = torch.randperm(NA), torch.randperm(NB)
    tX, tY = [(GA, lA, torch.argsort(lA))], [(GB, lB, torch.argsort(lB))]
    ref = dc.loss_fun_delta_cross_entropy(torch.bmm(FA, FB.transpose(1, 2)), tX, tY)
    ref.backward()
    gA, gB = FA.grad.clone(), FB.grad.clone()
    FA.grad = FB.grad = None
    got = dc.streamed_delta_cross_entropy(FA, FB, tX, tY, block=100)
    got.backward()
    assert abs(got.item() - ref.item()) < 1e-12
    assert torch.allclose(FA.grad, gA, atol=1e-12) and torch.allclose(FB.grad, gB, atol=1e-12)


def test_epoch_schedules_of_the_drivers():
    """as_rigid_as_possible/main.py:237-239 and mesh_mnist/main.py:174-176: lr *= 0.5 when epoch > 50 (20) and epoch % 10 == 0."""
    from surfacenetworks_amd import arap, mesh_mnist

    w = torch.nn.Parameter(torch.zeros(3))
    opt = torch.optim.Adam([w], 1e-3, weight_decay=1e-5)
    fired = [e for e in range(1, 101) if arap.halve_lr(opt, e)]
    assert fired == [60, 70, 80, 90, 100] and abs(opt.param_groups[0]["lr"] - 1e-3 / 32) < 1e-12
    opt = torch.optim.Adam([w], 1e-3, weight_decay=1e-5)
    fired = [e for e in range(1, 51) if mesh_mnist.halve_lr(opt, e)]
    assert fired == [30, 40, 50] and abs(opt.param_groups[0]["lr"] - 1e-3 / 8) < 1e-12
