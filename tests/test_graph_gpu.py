"""hipGraph replay of a training step == the same step launched eagerly, bit for bit (graphs.GraphedStep)."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _setup(model_kind, meshes=3, grid=(9, 8)):
    from surfacenetworks_amd import arap

    torch.manual_seed(5)
    ds = arap.ClothSequences([grid] * meshes, frames=arap.INPUT_FRAMES + arap.OUTPUT_FRAMES + 3, op_frames=3, seed=11,
                             device=DEV, model=model_kind)
    model = (arap.DirModel() if model_kind == "dir" else arap.Model()).to(DEV).train()
    return arap, ds, model


@pytest.mark.parametrize("model_kind", ["dir", "lap"])
def test_graph_replay_is_bit_identical_to_eager(model_kind):
    arap, ds, model_e = _setup(model_kind)
    _replay_equals_eager(arap, ds, model_e, np.arange(ds.n))


def test_graph_replay_of_the_headline_batch_is_bit_identical_to_eager():
    """BASELINE config 3 at its full size: 64 meshes of 71 x 71 per step, launch plans inside the captured step."""
    arap, ds, model_e = _setup("dir", 4, (71, 71))
    _replay_equals_eager(arap, ds, model_e, np.arange(64) % 4)


def _replay_equals_eager(arap, ds, model_e, ids):
    model_g = copy.deepcopy(model_e)
    opt_e, opt_g = arap.make_optimizer(model_e), arap.make_optimizer(model_g)
    n = len(ids)
    example = ds.sample_batch(n, np.random.default_rng(0), seq_ids=ids)
    state0 = [b.clone() for b in model_g.buffers()]
    graphed = arap.GraphedTrainStep(model_g, opt_g, example, global_batch=n)
    # capturing (warm-up + record) must leave the model state untouched
    for b0, b in zip(state0, model_g.buffers()):
        assert torch.equal(b0, b)
    rng_e, rng_g = np.random.default_rng(1), np.random.default_rng(1)
    for step in range(3):
        be = ds.sample_batch(n, rng_e, seq_ids=ids)
        bg = ds.sample_batch(n, rng_g, seq_ids=ids)
        assert graphed.matches(bg)
        le = arap.train_step(model_e, opt_e, be, global_batch=n)
        lg = graphed(bg)
        assert torch.equal(le.detach(), lg.detach()), f"loss differs at step {step}"
        for (name, pe), pg in zip(model_e.named_parameters(), model_g.parameters()):
            assert torch.equal(pe.grad, pg.grad), f"grad of {name} differs at step {step}"
            assert torch.equal(pe.detach(), pg.detach()), f"{name} differs after step {step}"
    for be_, bg_ in zip(model_e.buffers(), model_g.buffers()):
        assert torch.equal(be_, bg_)


def test_graph_replay_of_ragged_batches_in_permuted_order():
    """Meshes of different sizes: a replay on the same meshes in another order has the captured signature, and every
    per-batch quantity — the per-mesh vertex counts of the global-average blocks included — must come from the batch that
    was loaded, not from the example the graph was captured on."""
    from surfacenetworks_amd import arap

    torch.manual_seed(6)
    grids = [(9, 8), (7, 6), (8, 8)]
    ds = arap.ClothSequences(grids, frames=arap.INPUT_FRAMES + arap.OUTPUT_FRAMES + 3, op_frames=3, seed=12, device=DEV,
                             model="dir")
    model_e = arap.DirModel().to(DEV).train()
    model_g = copy.deepcopy(model_e)
    opt_e, opt_g = arap.make_optimizer(model_e), arap.make_optimizer(model_g)
    offs = np.zeros(3, dtype=np.int64)
    example = ds.sample_batch(3, None, seq_ids=np.array([0, 1, 2]), offsets=offs)
    graphed = arap.GraphedTrainStep(model_g, opt_g, example, global_batch=3)
    for ids in ([2, 0, 1], [1, 2, 0], [0, 1, 2]):
        be = ds.sample_batch(3, None, seq_ids=np.array(ids), offsets=offs)
        bg = ds.sample_batch(3, None, seq_ids=np.array(ids), offsets=offs)
        assert graphed.matches(bg)
        le = arap.train_step(model_e, opt_e, be, global_batch=3)
        lg = graphed(bg)
        assert torch.equal(le.detach(), lg.detach()), f"loss differs for order {ids}"
        for (name, pe), pg in zip(model_e.named_parameters(), model_g.parameters()):
            assert torch.equal(pe.detach(), pg.detach()), f"{name} differs after order {ids}"


def test_signature_mismatch_is_refused():
    arap, ds, model = _setup("dir", meshes=2)
    opt = arap.make_optimizer(model)
    graphed = arap.GraphedTrainStep(model, opt, ds.sample_batch(2, np.random.default_rng(0), seq_ids=np.arange(2)),
                                    global_batch=2)
    other = ds.sample_batch(1, np.random.default_rng(0), seq_ids=np.arange(1))
    assert not graphed.matches(other)
    with pytest.raises(ValueError):
        graphed(other)


def test_graph_replay_of_the_siamese_pair_step():
    """dense_correspondence: a batch type with its own tensor layout (PairBatch.graph_tensors).  The score matrix and the
    cross entropy go through torch / hipBLASLt, whose kernel choice may differ under capture, so: equal to rounding."""
    from surfacenetworks_amd import dense_correspondence as dc

    torch.manual_seed(3)
    ds = dc.TorusBodies(3, n=8, m=9, pad_to=80, seed=4, device=DEV)
    model_e = dc.SiameseModel("lap", 15).to(DEV).train()
    model_g = copy.deepcopy(model_e)
    opt_e, opt_g = dc.make_optimizer(model_e), dc.make_optimizer(model_g)
    graphed = dc.graphed_train_step(model_g, opt_g, dc.PairBatch(ds, 0, 1))
    for ia, ib in [(1, 2), (2, 0)]:
        le = dc.train_step(model_e, opt_e, ds, ia, ib)
        pb = dc.PairBatch(ds, ia, ib)
        assert graphed.matches(pb)
        lg = graphed(pb)
        assert torch.allclose(le.detach(), lg.detach(), rtol=1e-5, atol=1e-6), (le.item(), lg.item())
        # Adam's first steps move every entry by ~lr * sign(g): where the two runs' gradients (equal to fp32 rounding) straddle
        # zero the entries part by up to 2 lr per step; everywhere else they agree to rounding
        worst, far, total = 0.0, 0, 0
        for pe, pg in zip(model_e.parameters(), model_g.parameters()):
            d = (pe.detach() - pg.detach()).abs()
            worst = max(worst, float(d.max()))
            far += int((d > 2e-5 + 1e-3 * pe.detach().abs()).sum())
            total += d.numel()
        assert worst <= 2.1e-3 * (1 + [(1, 2), (2, 0)].index((ia, ib))) and far <= 2e-3 * total, (worst, far, total)


def test_batches_assembled_one_step_ahead_train_the_same_model():
    """graphs.BatchAhead (sampling + assembly on a side stream, one batch ahead of the consumer): the same sequence of
    batches, the same losses and parameters bit for bit as sampling on the compute stream — eager and replayed."""
    from surfacenetworks_amd import mesh_mnist as mm
    from surfacenetworks_amd.graphs import BatchAhead

    B = 48
    ds = mm.MeshDigits(B, seed=5, device=DEV, fixed_vertices=60, model="dir")
    ids = np.arange(B)
    torch.manual_seed(9)
    base = mm.DirModel().to(DEV).train()
    losses, params = [], []
    for mode in ("serial", "ahead", "ahead+graph"):
        model = copy.deepcopy(base)
        opt = mm.make_optimizer(model)
        rng = np.random.default_rng(11)
        make = lambda: ds.sample_batch(B, rng, ids=ids)       # noqa: E731
        torch.manual_seed(21)                                 # (dropout)
        if mode == "serial":
            out = [mm.train_step(model, opt, make()).detach().clone() for _ in range(6)]
        else:
            ahead = BatchAhead(make, DEV)
            step = mm.graphed_train_step(model, opt, ahead.get()) if mode.endswith("graph") else (lambda b: mm.train_step(model, opt, b))
            n = 5 if mode.endswith("graph") else 6            # (the capture consumed the first batch of the sequence)
            out = [step(ahead.get()).detach().clone() for _ in range(n)]
        torch.cuda.synchronize()
        losses.append(torch.stack([o.reshape(()) for o in out]).cpu())
        params.append([p.detach().clone() for p in model.parameters()])
    assert torch.equal(losses[0], losses[1]), (losses[0], losses[1])
    assert all(torch.equal(a, b) for a, b in zip(params[0], params[1]))
    # the replayed run skipped batch 0 (used as the capture example) and its dropout masks differ: finite, same batches' scale
    assert torch.isfinite(losses[2]).all()


def test_the_pair_step_issues_no_library_matrix_product():
    """One FAUST-style pair step (towers, correspondence loss, backward) on the device: every matrix product is one of the
    package's own kernels — no aten::mm / bmm / addmm / matmul reaches the library GEMM (models.py:203 and main.py:238-239 are
    sn_pair_fused_*; the towers' Linear layers the fused BatchNorm+Linear kernels)."""
    from torch.profiler import ProfilerActivity, profile

    from surfacenetworks_amd import dense_correspondence as dc

    torch.manual_seed(5)
    ds = dc.TorusBodies(2, n=8, m=9, pad_to=80, seed=4, device=DEV)
    model = dc.SiameseModel("lap", 15).to(DEV).train()
    batch = dc.PairBatch(ds, 0, 1)
    dc.forward_loss(model, batch).sum().backward()            # (lazy initialisations outside the profile)
    model.zero_grad(set_to_none=True)
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        loss = dc.forward_loss(model, dc.PairBatch(ds, 1, 0)).sum()
        loss.backward()
        torch.cuda.synchronize()
    names = {ev.name for ev in prof.events()}
    assert not names & {"aten::mm", "aten::bmm", "aten::addmm", "aten::matmul", "aten::baddbmm", "aten::linear"}, sorted(
        n for n in names if "mm" in n or "linear" in n or "matmul" in n)
    assert torch.isfinite(loss).item()


@pytest.mark.parametrize("kind", ["dir", "lap"])
def test_the_arap_step_issues_no_library_matrix_product(kind):
    """The ARAP training step (Dirac / Laplacian model, 120-output last layer, 6-channel first layer included) on the device:
    no aten::mm / addmm / bmm / linear — every Linear layer, weight gradient and sparse product is one of the package's kernels."""
    from torch.profiler import ProfilerActivity, profile

    from surfacenetworks_amd import arap

    torch.manual_seed(2)
    ds = arap.ClothSequences([(9, 8)] * 4, frames=arap.INPUT_FRAMES + arap.OUTPUT_FRAMES + 2, op_frames=2, seed=3, device=DEV, model=kind)
    model = (arap.DirModel() if kind == "dir" else arap.Model(15)).to(DEV).train()
    opt = arap.make_optimizer(model)
    rng = np.random.default_rng(1)
    arap.train_step(model, opt, ds.sample_batch(4, rng))
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        loss = arap.train_step(model, opt, ds.sample_batch(4, rng))
        torch.cuda.synchronize()
    names = {ev.name for ev in prof.events()}
    assert not names & {"aten::mm", "aten::bmm", "aten::addmm", "aten::matmul", "aten::baddbmm", "aten::linear"}, sorted(
        n for n in names if "mm" in n or "linear" in n or "matmul" in n)
    assert torch.isfinite(loss).item()


def test_the_samplers_cached_mask_is_never_a_graph_buffer():
    """ClothSequences keeps the masks of its last selections and hands the same tensor out when a selection recurs; the
    example batch of a capture must not make that tensor a static graph input (every load() would overwrite it, and the next
    batch of the example's selection would carry the previous batch's mask).  Ragged meshes, the example's selection coming
    back after another one: the replay equals an eager step on an INDEPENDENTLY built mask."""
    from surfacenetworks_amd import arap

    torch.manual_seed(8)
    grids = [(9, 8), (7, 6), (8, 8)]
    ds = arap.ClothSequences(grids, frames=arap.INPUT_FRAMES + arap.OUTPUT_FRAMES + 3, op_frames=3, seed=13, device=DEV,
                             model="dir")
    model_e = arap.DirModel().to(DEV).train()
    model_g = copy.deepcopy(model_e)
    opt_e, opt_g = arap.make_optimizer(model_e), arap.make_optimizer(model_g)
    offs = np.zeros(3, dtype=np.int64)
    first = np.array([0, 1, 2])
    example = ds.sample_batch(3, None, seq_ids=first, offsets=offs)
    cached = example.mask
    graphed = arap.GraphedTrainStep(model_g, opt_g, example, global_batch=3)
    assert graphed._g.step.static.mask is not cached
    nv = int(ds.num_vertices.max())

    def fresh_mask(ids):
        cnt = torch.from_numpy(ds.num_vertices[ids]).to(DEV)
        return (torch.arange(nv, device=DEV)[None, :] < cnt[:, None]).float().unsqueeze(2)

    for ids in ([2, 0, 1], [0, 1, 2], [1, 2, 0], [0, 1, 2]):
        ids = np.array(ids)
        bg = ds.sample_batch(3, None, seq_ids=ids, offsets=offs)
        assert torch.equal(bg.mask, fresh_mask(ids)), f"the sampler handed out a stale mask for {ids}"
        be = ds.sample_batch(3, None, seq_ids=ids, offsets=offs)
        be.mask = fresh_mask(ids)
        le = arap.train_step(model_e, opt_e, be, global_batch=3)
        lg = graphed(bg)
        assert torch.equal(le.detach(), lg.detach()), f"loss differs for selection {ids}"
        for (name, pe), pg in zip(model_e.named_parameters(), model_g.parameters()):
            assert torch.equal(pe.detach(), pg.detach()), f"{name} differs after selection {ids}"
