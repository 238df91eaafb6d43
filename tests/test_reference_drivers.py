"""`north_star`: "the existing mesh_mnist / as_rigid_as_possible / dense_correspondence training scripts run unmodified apart
from an import swap".  tests/test_import_swap.py runs the reference's three models.py files on the swapped operator layer; here
the DRIVERS' own function bodies run on it as well:

    src/as_rigid_as_possible/main.py   load_file :58-74, sample_batch :98-185, one iteration of the training loop :217-232
    src/mesh_mnist/main.py             convert :54-60, sample_batch :79-117, one iteration of the training loop :151-167
    src/dense_correspondence/main.py   read_data :66-102, sample_batch :106-191, loss_fun_delta_cross_entropy :229-240,
                                       one iteration of the training loop :310-327

The main.py modules cannot be imported (module-level argparse, dataset directory listings, `gcn` / `plyfile` / `seism`
imports), so — as tests/golden/make_dataset_fixtures.py does for the preprocessing functions — each function is taken from the
read-only checkout with `ast` AT TEST TIME, compiled from its own text and executed unmodified; nothing of it is stored here.
What the test supplies is the environment a driver run would have: `utils` = surfacenetworks_amd.utils_pt (the import swap),
`model` = the reference's models.py class built on the swapped layer, `args`, the data files tests/golden/data_* (written by the
reference's own preprocessing), and stand-ins for what this image lacks — a CUDA device (`.cuda()` is the identity,
torch.cuda.FloatTensor = torch.FloatTensor: the kernels behind the swapped layer are the oracle-backed host twins of the CPU
suite) and the third-party `seism` Delaunay Laplacian (the frame's stored `L`, the alternative the reference itself has commented
out at main.py:89).

Checked: (1) the reference sampler's batch == the product sampler's batch for the same draws (padded tensors equal, operators
equal); (2) loss and every parameter gradient of the reference's loop iteration == the product's own harness (its model class with
the same weights, its loss, its train step) on the same samples.  Skipped where /root/reference is absent."""
import ast
import os
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

REF_SRC = "/root/reference/src"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="reference checkout not present (it never travels)")


def _tree(rel):
    path = os.path.join(REF_SRC, rel)
    with open(path) as fh:
        return ast.parse(fh.read(), filename=path), path


def ref_function(rel, name, ns):
    """The module-level function `name` of a reference source file, compiled from its own text and bound to `ns`."""
    tree, path = _tree(rel)
    node = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)
    exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
    return ns[name]


def ref_train_iteration(rel, ns):
    """One iteration of the driver's training loop: the body of the first `for j in ...` loop nested in main()'s epoch loop,
    compiled from the reference's text as the body of a function of no arguments that returns its locals."""
    tree, path = _tree(rel)
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    epoch = next(n for n in ast.walk(main) if isinstance(n, ast.For) and getattr(n.target, "id", "") == "epoch")
    loop = next(n for n in epoch.body if isinstance(n, ast.For) and getattr(n.target, "id", "") == "j")
    ret = ast.parse("return locals()").body[0]
    fn = ast.FunctionDef(name="_iteration", args=ast.arguments(posonlyargs=[], args=[], kwonlyargs=[], kw_defaults=[], defaults=[]),
                         body=[ast.parse("loss_value = 0.0").body[0], ast.parse("correct = 0.0").body[0], *loop.body, ret],
                         decorator_list=[])
    mod = ast.fix_missing_locations(ast.Module(body=[fn], type_ignores=[]))
    exec(compile(mod, path, "exec"), ns)
    return ns["_iteration"]


def _grads(model):
    return {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def _same_weights(dst, src):
    dst.load_state_dict(src.state_dict())
    return dst


def _dense(op):
    return op.to_dense() if isinstance(op, torch.Tensor) else torch.from_numpy(op.to_scipy().toarray())


# ---- as_rigid_as_possible -----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["dir", "lap"])
def test_arap_driver_functions_run_on_the_swapped_layer(golden_dir, reference_models, kind):
    import surfacenetworks_amd.utils_pt as U
    from helpers import deterministic_init
    from surfacenetworks_amd import arap, datasets

    rel = "as_rigid_as_possible/main.py"
    args = types.SimpleNamespace(batch_size=3, model=kind, dense=False, cuda=False, num_updates=1, layer=15)
    files = [os.path.join(golden_dir, f"data_arap_seq{i}.npy") for i in (0, 1)]
    ns = {"np": np, "torch": torch, "utils": U, "args": args, "F": F, "test_ind": 0}
    load_file = ref_function(rel, "load_file", ns)
    # ten sequences (the reference trains on the first len // 10 * 8 of them): the two fixture files, repeated
    sequences = [load_file(files[i % 2]) for i in range(10)]
    sample_batch = ref_function(rel, "sample_batch", ns)
    models = reference_models["as_rigid_as_possible"]
    model = deterministic_init(models.DirModel() if kind == "dir" else models.Model(15), 7).train()
    opt = torch.optim.Adam(model.parameters(), 1e-3, weight_decay=1e-5)
    ns.update(sequences=sequences, sample_batch=sample_batch, model=model, early_optimizer=opt, tqdm=types.SimpleNamespace(tqdm=lambda x: x))
    iteration = ref_train_iteration(rel, ns)

    # (1) the reference sampler against the product's, same draws
    np.random.seed(11)
    seq_ids, offsets = [], []
    for _ in range(args.batch_size):
        ind = np.random.randint(0, len(sequences) // 10 * 8)
        seq_ids.append(ind)
        offsets.append(np.random.randint(0, len(sequences[ind]) - 2 - 40))
    np.random.seed(11)
    inputs, targets, mask, lap, Di, DiA, faces = sample_batch(sequences, True)
    ds = datasets.arap_from_files(files, device="cpu", model=kind, reorder=False)          # (the reference sampler hands out the file numbering)
    b = ds.sample_batch(args.batch_size, None, seq_ids=np.array(seq_ids) % 2, offsets=np.array(offsets))
    assert torch.equal(b.inputs, inputs) and torch.equal(b.targets, targets) and torch.equal(b.mask, mask)
    if kind == "dir":
        assert torch.equal(_dense(b.Di), _dense(Di)) and torch.equal(_dense(b.DiA), _dense(DiA))
    else:
        assert torch.equal(_dense(b.L), _dense(lap))

    # (2) one iteration of the reference's loop (its sample_batch, its model file, its loss, Adam) ...
    own = _same_weights(arap.DirModel() if kind == "dir" else arap.Model(15), model).train()      # (before the update)
    np.random.seed(11)
    out = iteration()
    ref_loss, ref_grads = out["loss"].detach(), _grads(model)
    # ... against the product's harness on the same samples: its model class, its fused loss, its train step
    own_opt = arap.make_optimizer(own)
    own_loss = arap.train_step(own, own_opt, b, global_batch=args.batch_size)
    assert abs(own_loss.item() - ref_loss.item()) <= 1e-5 * abs(ref_loss.item())
    own_grads = _grads(own)
    assert own_grads.keys() == ref_grads.keys()
    worst = max(_rel(own_grads[k], ref_grads[k]) for k in ref_grads)
    assert worst < 2e-4, worst                # (two fp32 evaluation orders of a 15-block model; the golden tests bound each against fp64)
    for (k, a), (_, r) in zip(own.state_dict().items(), model.state_dict().items()):
        assert torch.allclose(a.float(), r.float(), rtol=1e-3, atol=1e-5), k      # the same Adam update


# ---- mesh_mnist -----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["lap", "dirac"])
def test_mesh_mnist_driver_functions_run_on_the_swapped_layer(golden_dir, reference_models, kind):
    import surfacenetworks_amd.utils_pt as U
    from helpers import deterministic_init
    from surfacenetworks_amd import datasets, mesh_mnist as mm
    from torch.autograd import Variable

    rel = "mesh_mnist/main.py"
    args = types.SimpleNamespace(batch_size=3, cuda=False, model=kind)
    with open(os.path.join(golden_dir, "data_mnist_plus.np"), "rb") as fh:
        raw = list(np.load(fh, encoding="latin1", allow_pickle=True))          # (main.py:49, with the flag numpy >= 1.16.3 wants)
    import copy

    train_data = copy.deepcopy(raw)
    ns = {"np": np, "torch": torch, "utils": U, "args": args, "F": F, "Variable": Variable,
          "inputs": torch.zeros(1, 1, 3), "targets": torch.zeros(1).long(), "mask": torch.zeros(1, 1, 1),
          "tqdm": types.SimpleNamespace(tqdm=lambda x: x), "gc": __import__("gc")}
    convert = ref_function(rel, "convert", ns)
    for s in train_data:
        convert(s)
    sample_batch = ref_function(rel, "sample_batch", ns)
    sample_batch.num_vertices = 0
    sample_batch.num_faces = 0
    models = reference_models["mesh_mnist"]
    model = deterministic_init(models.Model() if kind == "lap" else models.DirModel(), 5).train()
    opt = torch.optim.Adam(model.parameters(), 1e-3, weight_decay=1e-5)
    ns.update(train_data=train_data, sample_batch=sample_batch, model=model, early_optimizer=opt)
    iteration = ref_train_iteration(rel, ns)

    np.random.seed(3)
    ids = [np.random.randint(0, len(train_data)) for _ in range(args.batch_size)]
    np.random.seed(3)
    inputs, targets, mask, lap, Di, DiA = sample_batch(train_data, is_training=True)
    # (reorder=False: the file's own vertex numbering, which is what the reference's sampler hands out; the default stores the
    #  meshes in a locality numbering — tests/test_order*.py)
    ds = datasets.mnist_from_samples(raw, device="cpu", model="lap" if kind == "lap" else "dir", reorder=False)
    b = ds.sample_batch(args.batch_size, None, ids=np.array(ids))
    assert torch.equal(b.inputs, inputs) and torch.equal(b.targets, targets) and torch.equal(b.mask, mask)
    nv, nf = sample_batch.num_vertices, sample_batch.num_faces
    if kind == "lap":
        assert torch.equal(_dense(b.L), torch.block_diag(*lap.to_dense()))
    else:
        assert torch.equal(_dense(b.Di), torch.block_diag(*Di.to_dense())) and tuple(Di.shape) == (args.batch_size, 4 * nf, 4 * nv)

    own = _same_weights(mm.Model() if kind == "lap" else mm.DirModel(), model).train()
    sample_batch.num_vertices = sample_batch.num_faces = 0               # (the running maxima of main.py:83-84 start over)
    np.random.seed(3)
    torch.manual_seed(17)                       # both models draw ONE dropout mask, of the same shape (models.py:50 / :156)
    out = iteration()
    ref_loss, ref_grads = out["loss"].detach(), _grads(model)
    own_opt = mm.make_optimizer(own)
    torch.manual_seed(17)
    own_loss = mm.train_step(own, own_opt, b)
    assert abs(own_loss.item() - ref_loss.item()) <= 2e-5 * abs(ref_loss.item())
    own_grads = _grads(own)
    assert own_grads.keys() == ref_grads.keys()
    worst = max(_rel(own_grads[k], ref_grads[k]) for k in ref_grads)
    assert worst < 5e-3, worst                # (cotangent operators with entries of 1e4 and rows summing to 0: DESIGN.md §5)


# ---- dense_correspondence --------------------------------------------------------------------------------------------------------
def test_faust_driver_functions_run_on_the_swapped_layer(golden_dir, reference_models, monkeypatch):
    import scipy as sp
    import scipy.sparse  # noqa: F401

    import surfacenetworks_amd.utils_pt as U
    from helpers import deterministic_init
    from surfacenetworks_amd import datasets, dense_correspondence as dc

    rel = "dense_correspondence/main.py"
    path = os.path.join(golden_dir, "data_faust_frame.npz")
    args = types.SimpleNamespace(batch_size=1, model="lap", layer=3, full_train=True, xz_rotate=False, xy_rotate=False, num_updates=1)
    # what this image lacks: a CUDA device, and `seism` behind utils.mesh.intrinsic_laplacian (main.py:88) — the frame's stored
    # Laplacian takes its place, the alternative the reference keeps commented out one line below
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self, raising=False)
    monkeypatch.setattr(torch.cuda, "FloatTensor", torch.FloatTensor, raising=False)
    with np.load(path, allow_pickle=True) as z:
        stored_L = z["L"].item().astype("f").tocsr()
    mesh = types.SimpleNamespace(intrinsic_laplacian=lambda V, Fc: stored_L)
    np_load = np.load
    npx = types.SimpleNamespace(**{k: getattr(np, k) for k in ("sqrt", "random", "cos", "sin", "pi")},
                                load=lambda p, *a, **k: np_load(p, *a, allow_pickle=True, **k))
    ns = {"np": npx, "torch": torch, "utils": U, "mesh": mesh, "sp": sp, "F": F, "gc": __import__("gc"), "random": __import__("random")}
    read_data = ref_function(rel, "read_data", ns)
    sample_batch = ref_function(rel, "sample_batch", ns)
    sample_batch.num_vertices, sample_batch.num_faces, sample_batch.test_ind = 64, 0, 0       # (main.py:193-195, pad_to of the fixture)
    loss_fun = ref_function(rel, "loss_fun_delta_cross_entropy", ns)
    sequences = [read_data(path, args) for _ in range(10)]          # (sample_batch draws from the first len // 10 * 8 before --full-train widens it)
    nv = int(sequences[0]["V"].shape[0])
    pad = max(64, nv)
    sample_batch.num_vertices = pad
    models = reference_models["dense_correspondence"]
    model = deterministic_init(models.SiameseModel("lap", 3), 9).train()
    opt = torch.optim.Adam(model.parameters(), 1e-3, weight_decay=1e-5)
    ns.update(sequences=sequences, sample_batch=sample_batch, model=model, early_optimizer=opt, loss_fun=loss_fun, args=args)
    iteration = ref_train_iteration(rel, ns)

    # (1) the reference's reader and sampler against the product's
    ds = datasets.faust_from_files([path, path], device="cpu", model="lap", pad_to=pad, reorder=False)
    inX, tX, mX, LX = ds.sample(0)
    rin, rt, rm, rop, _ = sample_batch(sequences, True, args)
    assert torch.equal(inX, rin) and torch.equal(mX, rm)
    assert torch.equal(tX[0][0], rt[0][0]) and torch.equal(tX[0][1], rt[0][1]) and torch.equal(tX[0][2], rt[0][2])
    assert torch.equal(_dense(LX), _dense(rop))

    # (2) one iteration of the reference's loop against the product's pair step on the same two samples
    own = _same_weights(dc.SiameseModel("lap", 3), model).train()
    out = iteration()
    ref_loss, ref_grads = out["loss"].detach(), _grads(model)
    own_opt = dc.make_optimizer(own)
    own_loss = dc.train_step(own, own_opt, ds, 0, 1)
    assert abs(own_loss.item() - ref_loss.item()) <= 1e-5 * abs(ref_loss.item())
    own_grads = _grads(own)
    assert own_grads.keys() == ref_grads.keys()
    worst = max(_rel(own_grads[k], ref_grads[k]) for k in ref_grads)
    assert worst < 1e-3, worst
