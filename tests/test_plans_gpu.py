"""Launch plans (surfacenetworks_amd/plans.py, csrc/sn_plan.hip): a block enqueued by one sn_plan_run is the block launched
kernel by kernel from Python — same kernels, same order, same arguments — so every result must be BIT-identical: losses,
parameter gradients, updated parameters and BatchNorm running statistics over several training steps of every model family,
padded and packed batches, two steps per shape (the recording call and a pure replay), changing operators per step."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True)
def _fresh_plans():
    from surfacenetworks_amd import plans

    plans.reset()
    plans.set_enabled(True)
    plans.set_graphs(True)
    yield
    plans.set_enabled(True)
    plans.set_graphs(True)


def _steps(step_fn, model_e, model_p, n_steps, check_no_refusal=True):
    """step_fn(model, k) -> loss runs one training step; eager on model_e (plans off), planned on model_p."""
    from surfacenetworks_amd import plans

    for k in range(n_steps):
        plans.set_enabled(False)
        torch.manual_seed(100 + k)                  # (dropout in the Mesh-MNIST head draws from the device generator)
        le = step_fn(model_e, k)
        plans.set_enabled(True)
        torch.manual_seed(100 + k)
        lp = step_fn(model_p, k)
        assert torch.equal(le.detach(), lp.detach()), f"loss differs at step {k}: {le.item()} vs {lp.item()}"
        for (name, pe), pp in zip(model_e.named_parameters(), model_p.parameters()):
            assert (pe.grad is None) == (pp.grad is None), name
            if pe.grad is not None:
                assert torch.equal(pe.grad, pp.grad), f"grad of {name} differs at step {k}"
            assert torch.equal(pe.detach(), pp.detach()), f"{name} differs after step {k}"
        for (name, be), bp in zip(model_e.named_buffers(), model_p.buffers()):
            assert torch.equal(be, bp), f"buffer {name} differs after step {k}"
    st = plans.stats()
    assert sum(s["replayed"] for s in st.values()) > 0, st
    from surfacenetworks_amd import kernels

    if check_no_refusal and kernels._split_gemm():    # (the fp32-MFMA A/B leg, SN_GEMM_VARIANT=0, sends the 120-wide layer to the library: refused)
        assert all(s["refused"] == 0 for s in st.values()), st
    return st


@pytest.mark.parametrize("model_kind", ["dir", "lap"])
def test_arap_steps_planned_equal_eager(model_kind):
    from surfacenetworks_amd import arap

    torch.manual_seed(5)
    ds = arap.ClothSequences([(9, 8)] * 3, frames=arap.INPUT_FRAMES + arap.OUTPUT_FRAMES + 3, op_frames=3, seed=11, device=DEV,
                             model=model_kind)
    model_e = (arap.DirModel() if model_kind == "dir" else arap.Model()).to(DEV).train()
    model_p = copy.deepcopy(model_e)
    opts = {id(model_e): arap.make_optimizer(model_e), id(model_p): arap.make_optimizer(model_p)}
    rngs = {id(model_e): np.random.default_rng(1), id(model_p): np.random.default_rng(1)}

    def step(model, k):
        batch = ds.sample_batch(3, rngs[id(model)], seq_ids=np.arange(3))
        return arap.train_step(model, opts[id(model)], batch, global_batch=3)

    st = _steps(step, model_e, model_p, 4)
    blk = "dirac" if model_kind == "dir" else "propagate"
    assert st[f"{blk}_fwd"]["replayed"] >= 4 * 8 and st[f"{blk}_bwd"]["replayed"] >= 4 * 8
    assert st["avg_fwd"]["replayed"] >= 4 * 7 and st["elu_conv_fwd"]["replayed"] >= 4
    # one plan per distinct signature, recorded once: first Dirac block (zero faces), middle blocks, last block (no tile sums)
    assert st[f"{blk}_fwd"]["recorded"] <= 3 and st["avg_fwd"]["recorded"] <= 2


def test_the_headline_batch_planned_equals_eager():
    """BASELINE config 3 at its full size (64 meshes of 71 x 71, the arenas in the hundreds of MB, the global-average stages past
    the merged launch's mesh limit): two steps, planned against eager, every gradient, parameter and buffer bit for bit."""
    from surfacenetworks_amd import arap

    torch.manual_seed(7)
    ds = arap.ClothSequences([(71, 71)] * 4, frames=arap.INPUT_FRAMES + arap.OUTPUT_FRAMES + 2, op_frames=2, seed=3, device=DEV, model="dir")
    model_e = arap.DirModel().to(DEV).train()
    model_p = copy.deepcopy(model_e)
    opts = {id(model_e): arap.make_optimizer(model_e), id(model_p): arap.make_optimizer(model_p)}
    rngs = {id(model_e): np.random.default_rng(2), id(model_p): np.random.default_rng(2)}

    def step(model, k):
        batch = ds.sample_batch(64, rngs[id(model)], seq_ids=np.arange(64) % 4)
        return arap.train_step(model, opts[id(model)], batch, global_batch=64)

    st = _steps(step, model_e, model_p, 2)
    assert st["dirac_fwd"]["replayed"] >= 2 * 8 and st["avg_bwd"]["replayed"] >= 2 * 7, st


def test_packed_ragged_batch_planned_equals_eager():
    from surfacenetworks_amd import arap

    torch.manual_seed(6)
    grids = [(9, 8), (7, 6), (8, 8)]
    ds = arap.ClothSequences(grids, frames=arap.INPUT_FRAMES + arap.OUTPUT_FRAMES + 3, op_frames=3, seed=12, device=DEV, model="dir")
    model_e = arap.DirModel().to(DEV).train()
    model_p = copy.deepcopy(model_e)
    opts = {id(model_e): arap.make_optimizer(model_e), id(model_p): arap.make_optimizer(model_p)}
    orders = [[0, 1, 2], [2, 0, 1], [0, 1, 2], [1, 2, 0]]

    def step(model, k):
        batch = ds.sample_batch(3, None, seq_ids=np.array(orders[k]), offsets=np.zeros(3, dtype=np.int64), packed=True)
        return arap.train_step(model, opts[id(model)], batch, global_batch=3)

    st = _steps(step, model_e, model_p, 4)
    assert st["avg_ragged_fwd"]["replayed"] >= 4 * 7, st


def test_mesh_mnist_dirac_planned_equals_eager():
    from surfacenetworks_amd import mesh_mnist as mm

    torch.manual_seed(7)
    ds = mm.MeshDigits(12, seed=2, device=DEV, fixed_vertices=150, model="dir")
    model_e = mm.DirModel().to(DEV).train()
    model_p = copy.deepcopy(model_e)
    opts = {id(model_e): mm.make_optimizer(model_e), id(model_p): mm.make_optimizer(model_p)}
    rngs = {id(model_e): np.random.default_rng(3), id(model_p): np.random.default_rng(3)}

    def step(model, k):
        return mm.train_step(model, opts[id(model)], ds.sample_batch(8, rngs[id(model)]))

    _steps(step, model_e, model_p, 3)


def test_faust_pair_planned_equals_eager():
    from surfacenetworks_amd import dense_correspondence as dc

    torch.manual_seed(8)
    ds = dc.TorusBodies(3, n=13, m=17, pad_to=256, seed=5, device=DEV)
    model_e = dc.SiameseModel("lap", 15).to(DEV).train()
    model_p = copy.deepcopy(model_e)
    opts = {id(model_e): dc.make_optimizer(model_e), id(model_p): dc.make_optimizer(model_p)}

    def step(model, k):
        return dc.train_step(model, opts[id(model)], ds, k % 3, (k + 1) % 3)

    _steps(step, model_e, model_p, 3)


def test_eval_mode_and_inference_fall_back_or_replay_identically():
    """Evaluation (running statistics, no backward): whatever the planned path does — replay or refuse — the outputs are those of
    the eager path."""
    from surfacenetworks_amd import arap, plans

    torch.manual_seed(9)
    ds = arap.ClothSequences([(9, 8)] * 2, frames=arap.INPUT_FRAMES + arap.OUTPUT_FRAMES + 3, op_frames=3, seed=13, device=DEV, model="dir")
    model = arap.DirModel().to(DEV).eval()
    batch = ds.sample_batch(2, np.random.default_rng(0), seq_ids=np.arange(2))
    with torch.no_grad():
        plans.set_enabled(False)
        want = model(batch.Di, batch.DiA, batch.mask, batch.inputs)
        plans.set_enabled(True)
        got1 = model(batch.Di, batch.DiA, batch.mask, batch.inputs)
        got2 = model(batch.Di, batch.DiA, batch.mask, batch.inputs)
    assert torch.equal(want, got1) and torch.equal(want, got2)


def test_plan_entry_table_is_the_launchers_of_the_header():
    import ctypes as C

    from surfacenetworks_amd import _lib

    lib = _lib.load()
    n = int(lib.sn_plan_entry_count())
    names = [lib.sn_plan_entry_name(i).decode() for i in range(n)]
    for i, name in enumerate(names):
        res, args = _lib.SIGNATURES[name]
        sig = lib.sn_plan_entry_signature(i).decode()
        want = "".join("p" if a is C.c_void_p else "d" if a in (C.c_double, C.c_float) else "i" for a in args)
        assert sig == want, (name, sig, want)
        assert int(lib.sn_plan_lookup(name.encode())) == i


def _one_block(C=128):
    from surfacenetworks_amd import mesh_ops
    from surfacenetworks_amd import utils_pt as U
    from surfacenetworks_amd.operators import SparseOperator

    rng = np.random.default_rng(3)
    V, F = mesh_ops.grid_cloth(12, 9, rng)
    ops = mesh_ops.mesh_operators(V, F)
    L = SparseOperator.from_scipy(ops["L"], DEV)
    torch.manual_seed(4)
    blk = U.LapResNet2(C).to(DEV).train()
    x = torch.randn(1, V.shape[0], C, device=DEV)
    return blk, L, x


def test_second_backward_through_a_planned_block_says_what_to_do():
    from surfacenetworks_amd import plans

    blk, L, x = _one_block()
    plans.set_enabled(True)
    xi = x.clone().requires_grad_(True)
    loss = blk(L, None, xi).sum()
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="SN_PLANS=0"):
        loss.backward()


def test_planned_forward_without_a_backward_releases_its_workspace():
    from surfacenetworks_amd import plans

    blk, L, x = _one_block()
    plans.set_enabled(True)
    with torch.no_grad():
        blk(L, None, x)                                            # records
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    base = torch.cuda.memory_allocated()
    for _ in range(5):
        with torch.no_grad():
            y = blk(L, None, x)
        del y
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() <= base + (1 << 20)       # nothing accumulates over calls


def test_a_hand_built_plan_through_the_c_abi_equals_the_two_calls_it_records():
    """The binding INTEGRATION.md §2 shows: an ELU pass and a CSR product recorded into a plan through ctypes alone, run on new
    tensors — against the same two entry points called one by one."""
    import ctypes

    from helpers import mesh_fixture
    from surfacenetworks_amd import _lib

    lib = _lib.load()
    _, _, ops = mesh_fixture("cloth")
    A = ops["L"].tocsr()
    A.sort_indices()
    rows, C, nnz = A.shape[0], 64, A.nnz
    rowptr = torch.from_numpy(A.indptr.astype(np.int32)).to(DEV)
    colind = torch.from_numpy(A.indices.astype(np.int32)).to(DEV)
    vals = torch.from_numpy(A.data.astype(np.float32)).to(DEV)
    plan = ctypes.c_void_p()
    assert lib.sn_plan_create(ctypes.byref(plan)) == 0

    def add(name, kinds, slots, ivals):
        n = len(kinds)
        I32, I64, F64 = ctypes.c_int32 * n, ctypes.c_int64 * n, ctypes.c_double * n
        st = lib.sn_plan_add_call(plan, lib.sn_plan_lookup(name), n, I32(*kinds), I32(*slots), I64(*ivals), F64())
        assert st == 0, lib.sn_status_string(st)

    add(b"sn_elu_into_f32", [2, 0, 2, 0, 0, 0, 4], [0, 0, 1, 0, 0, 0, 0], [0, C, 0, 2 * C, rows, C, 0])
    add(b"sn_spmm_csr_f32", [2, 2, 2, 0, 0, 0, 2, 0, 0, 0, 2, 0, 0, 4], [2, 3, 4, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0],
        [0, 0, 0, rows, rows, nnz, 0, 2 * C, 1, C, 4 * C, 2 * C, 1, 0])
    assert lib.sn_plan_length(plan) == 2
    stream = torch.cuda.current_stream().cuda_stream
    for seed in (0, 1):                                            # the same plan on new tensors
        g = torch.Generator(device=DEV).manual_seed(seed)
        x = torch.randn(rows, C, device=DEV, generator=g)
        y = torch.full((rows, 2 * C), float("nan"), device=DEV)
        bases = (ctypes.c_uint64 * 5)(x.data_ptr(), y.data_ptr(), rowptr.data_ptr(), colind.data_ptr(), vals.data_ptr())
        assert lib.sn_plan_run(plan, bases, 5, stream, None) == 0
        want = torch.full((rows, 2 * C), float("nan"), device=DEV)
        _lib.call("sn_elu_into_f32", x.data_ptr(), C, want.data_ptr(), 2 * C, rows, C, stream)
        _lib.call("sn_spmm_csr_f32", rowptr.data_ptr(), colind.data_ptr(), vals.data_ptr(), rows, rows, nnz, want.data_ptr(), 2 * C, 1, C,
                  want.data_ptr() + 4 * C, 2 * C, 1, stream)
        torch.cuda.synchronize()
        assert torch.equal(y, want) and not torch.isnan(y).any()
    assert lib.sn_plan_destroy(plan) == 0


def test_per_launch_timer_sees_the_same_products_through_plans():
    """functional.SpmmTimer (bench.py's roofline measurement) under launch plans: the plan replays the tags of its sparse
    products into the active timer, the C-side records come from the same launchers — same list as the eager step's."""
    from surfacenetworks_amd import arap, functional as snF, plans

    torch.manual_seed(5)
    ds = arap.ClothSequences([(9, 8)] * 3, frames=arap.INPUT_FRAMES + arap.OUTPUT_FRAMES + 3, op_frames=3, seed=11, device=DEV, model="dir")
    model = arap.DirModel().to(DEV).train()
    opt = arap.make_optimizer(model)
    got = {}
    for on in (False, True):
        plans.set_enabled(on)
        for _ in range(2):                                         # (with plans: the recording step, then a pure replay)
            arap.train_step(model, opt, ds.sample_batch(3, np.random.default_rng(1), seq_ids=np.arange(3)), global_batch=3)
        timer = snF.SpmmTimer()
        with timer:
            arap.train_step(model, opt, ds.sample_batch(3, np.random.default_rng(1), seq_ids=np.arange(3)), global_batch=3)
        recs = timer.results()
        got[on] = ([r[:5] for r in recs], sorted((k, r, w, o) for k, r, w, o, _b, _ms in timer.linear))
        assert all(r[5] > 0 for r in recs)
    assert got[True] == got[False] and len(got[True][0]) == 32     # 8 Dirac blocks x (2 forward + 2 backward products)
    assert plans.stats()["dirac_fwd"]["replayed"] >= 16


def test_a_block_whose_plan_is_refused_runs_exactly_the_eager_path(monkeypatch):
    """A refused (or cooling-down) plan must leave nothing behind: the eager fallback still finds BatchNorm's one-shot batch
    counter on the running-mean buffer (found with SN_GEMM_VARIANT=0, where the last layer's plan is refused: the counter
    stayed at 0)."""
    from surfacenetworks_amd import plans

    blk, L, x = _one_block()
    plans.set_enabled(True)
    monkeypatch.setattr(plans, "record", lambda site, key, *a, **k: site.refuse(key, "refused by the test"))
    for _ in range(3):
        blk(L, None, x.clone().requires_grad_(True)).sum().backward()
    st = plans.stats()
    assert st["propagate_fwd"]["refused"] == 1 and st["propagate_fwd"]["replayed"] == 0
    assert int(blk.bn_fc0.bn.num_batches_tracked) == 3 and int(blk.bn_fc1.bn.num_batches_tracked) == 3


def test_the_placeholder_face_features_are_never_materialised():
    """need_f=False hands the next Dirac block a zero-stride NaN placeholder as `f` (the activated hand-off carries the values):
    nothing may make it contiguous — a (faces, C) copy per block, 321 MB at the ARAP batch (a round-6 regression, found by a
    same-box A/B against the round-5 tree: +0.33 ms per step)."""
    from surfacenetworks_amd import arap, plans

    torch.manual_seed(5)
    ds = arap.ClothSequences([(9, 8)] * 2, frames=arap.INPUT_FRAMES + arap.OUTPUT_FRAMES + 3, op_frames=3, seed=11, device=DEV, model="dir")
    model = arap.DirModel().to(DEV).train()
    opt = arap.make_optimizer(model)
    batch = ds.sample_batch(2, np.random.default_rng(1), seq_ids=np.arange(2))
    arap.train_step(model, opt, batch, global_batch=2)
    orig = torch.Tensor.contiguous
    copies = []

    def spy(self, *a, **k):
        if not self.is_contiguous() and self.dim() == 2 and self.stride() == (0, 0):
            copies.append(tuple(self.shape))
        return orig(self, *a, **k)

    torch.Tensor.contiguous = spy
    try:
        for on in (True, False):
            plans.set_enabled(on)
            arap.train_step(model, opt, ds.sample_batch(2, np.random.default_rng(1), seq_ids=np.arange(2)), global_batch=2)
    finally:
        torch.Tensor.contiguous = orig
        plans.set_enabled(True)
    assert copies == []


@pytest.mark.parametrize("off,pitch,width,rows", [(0, 1024, 512, 37), (4, 1028, 516, 9), (3, 611, 333, 5), (16, 0, 1 << 20, 1), (1, 0, 7, 1),
                                                    (0, 64, 64, 1000), (0, 0, 0, 1)])
def test_plan_fill_and_copy_nodes_write_exactly_their_region(off, pitch, width, rows):
    """The fill / copy nodes are kernels of the library (16-, 4- or 1-byte units by alignment): the region and nothing around it."""
    import ctypes

    from surfacenetworks_amd import _lib

    lib = _lib.load()
    span = off + (pitch * (rows - 1) if rows > 1 else 0) + width + 64
    g = torch.Generator(device=DEV).manual_seed(5)
    src = torch.randint(1, 255, (span,), dtype=torch.uint8, device=DEV, generator=g)
    dst_f = torch.full((span,), 0xAA, dtype=torch.uint8, device=DEV)
    dst_c = dst_f.clone()
    plan = ctypes.c_void_p()
    assert lib.sn_plan_create(ctypes.byref(plan)) == 0
    assert lib.sn_plan_add_memset(plan, 0, off, 0x5C, pitch, width, rows) == 0
    assert lib.sn_plan_add_copy(plan, 1, off, pitch, 2, off, pitch, width, rows) == 0
    bases = (ctypes.c_uint64 * 3)(dst_f.data_ptr(), dst_c.data_ptr(), src.data_ptr())
    assert lib.sn_plan_run(plan, bases, 3, torch.cuda.current_stream().cuda_stream, None) == 0
    torch.cuda.synchronize()
    want_f = torch.full((span,), 0xAA, dtype=torch.uint8)
    want_c = want_f.clone()
    s = src.cpu()
    for r in range(rows):
        a = off + r * pitch
        want_f[a:a + width] = 0x5C
        want_c[a:a + width] = s[a:a + width]
    assert torch.equal(dst_f.cpu(), want_f)
    assert torch.equal(dst_c.cpu(), want_c)
    lib.sn_plan_destroy(plan)


def test_plan_fill_nodes_replay_correctly_from_a_captured_graph():
    """A flat hipMemsetAsync captured into a graph replays with wrong bytes from the second replay on (ROCm 7.2): the plan's fills
    are kernels, and a plan captured by a caller's graph gives the eager bytes on every replay."""
    import ctypes

    from surfacenetworks_amd import _lib

    lib = _lib.load()
    rows, C = 128, 256
    for twod in (False, True):
        plan = ctypes.c_void_p()
        assert lib.sn_plan_create(ctypes.byref(plan)) == 0
        if twod:
            assert lib.sn_plan_add_memset(plan, 0, 0, 0, C * 4, C // 2 * 4, rows) == 0
        else:
            assert lib.sn_plan_add_memset(plan, 0, 0, 0, rows * C * 4, rows * C * 4, 1) == 0
        n = 7
        I32, I64, F64 = ctypes.c_int32 * n, ctypes.c_int64 * n, ctypes.c_double * n
        assert lib.sn_plan_add_call(plan, lib.sn_plan_lookup(b"sn_elu_into_f32"), n, I32(2, 0, 2, 0, 0, 0, 4), I32(0, 0, 1, 0, 0, 0, 0),
                                    I64(0, C, 0, C, rows, C, 0), F64()) == 0
        x = torch.full((rows, C), 3.0, device=DEV)
        y = torch.empty((rows, C), device=DEV)
        bases = (ctypes.c_uint64 * 2)(x.data_ptr(), y.data_ptr())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            assert lib.sn_plan_run(plan, bases, 2, torch.cuda.current_stream().cuda_stream, None) == 0
        for rep in range(4):
            x.fill_(5.0 + rep)
            y.fill_(-1.0)
            graph.replay()
            torch.cuda.synchronize()
            zero_cols = C // 2 if twod else C
            assert float(y[:, :zero_cols].abs().max()) == 0.0, (twod, rep)
            if twod:
                assert torch.equal(y[:, zero_cols:], torch.full((rows, C - zero_cols), 5.0 + rep, device=DEV))
        del graph
        lib.sn_plan_destroy(plan)


def test_plan_runs_at_repeating_addresses_are_graph_launches_with_the_same_results(monkeypatch):
    """In a training loop the slot addresses of a plan run repeat (the caching allocator); from the second sighting such a run is ONE
    graph launch (sn_plan_instantiate / sn_plan_exec_launch).  Same kernels, same arguments: the model trained with graph launches
    equals, bit for bit, the one whose plan runs walk their lists — and plans past the workspace limit never use a graph."""
    from surfacenetworks_amd import arap, plans

    torch.manual_seed(5)
    ds = arap.ClothSequences([(9, 8)] * 3, frames=arap.INPUT_FRAMES + arap.OUTPUT_FRAMES + 3, op_frames=3, seed=11, device=DEV, model="dir")
    model_w = arap.DirModel().to(DEV).train()
    model_g = copy.deepcopy(model_w)
    monkeypatch.setattr(plans, "GRAPH_AFTER_SIGHTINGS", 2)        # (the default buys a graph at the twelfth sighting of an address set)
    for model, graphs in ((model_w, False), (model_g, True)):
        plans.reset()
        plans.set_graphs(graphs)
        opt = arap.make_optimizer(model)
        rng = np.random.default_rng(1)
        for _ in range(8):
            arap.train_step(model, opt, ds.sample_batch(3, rng, seq_ids=np.arange(3)), global_batch=3)
        torch.cuda.synchronize()
        g = plans.graph_stats()
        if graphs:
            assert g["instantiated"] >= 16 and g["launched"] >= 5 * g["instantiated"] and g["refused"] == 0, g
        else:
            assert g["launched"] == g["instantiated"] == 0, g
    for (name, pw), pg in zip(model_w.named_parameters(), model_g.parameters()):
        assert torch.equal(pw.detach(), pg.detach()), name
    for (name, bw), bg in zip(model_w.named_buffers(), model_g.buffers()):
        assert torch.equal(bw, bg), name
    monkeypatch.setattr(plans, "GRAPH_MAX_ARENA_BYTES", 1024)
    plans.reset()
    opt = arap.make_optimizer(model_g)
    for _ in range(4):
        arap.train_step(model_g, opt, ds.sample_batch(3, np.random.default_rng(2), seq_ids=np.arange(3)), global_batch=3)
    assert plans.graph_stats()["instantiated"] == 0 and sum(s["replayed"] for s in plans.stats().values()) > 0


def test_a_plan_as_a_graph_through_the_c_abi():
    """sn_plan_instantiate at fixed addresses, then sn_plan_exec_launch: the same bytes as sn_plan_run on new CONTENTS of the same
    buffers; inside a capture of the caller's it walks the list (and the caller's graph replays right); refused while the
    per-launch timer is on."""
    import ctypes

    from surfacenetworks_amd import _lib

    lib = _lib.load()
    rows, C = 257, 128
    plan = ctypes.c_void_p()
    assert lib.sn_plan_create(ctypes.byref(plan)) == 0
    assert lib.sn_plan_add_memset(plan, 1, 0, 0, C * 4, C // 2 * 4, rows) == 0                      # zero the left half of y
    n = 7
    I32, I64, F64 = ctypes.c_int32 * n, ctypes.c_int64 * n, ctypes.c_double * n
    assert lib.sn_plan_add_call(plan, lib.sn_plan_lookup(b"sn_elu_into_f32"), n, I32(2, 0, 2, 0, 0, 0, 4), I32(0, 0, 1, 0, 0, 0, 0),
                                I64(0, C, C // 2 * 4, C, rows, C // 2, 0), F64()) == 0             # elu(x[:, :C/2]) into the right half
    x = torch.randn(rows, C, device=DEV)
    y = torch.empty(rows, C, device=DEV)
    bases = (ctypes.c_uint64 * 2)(x.data_ptr(), y.data_ptr())
    ex = ctypes.c_void_p()
    assert lib.sn_plan_instantiate(plan, bases, 2, ctypes.byref(ex), None) == 0 and ex
    stream = torch.cuda.current_stream().cuda_stream

    def want():
        return torch.cat([torch.zeros(rows, C // 2, device=DEV), torch.nn.functional.elu(x[:, :C // 2])], dim=1)

    for seed in range(3):
        x.copy_(torch.randn(rows, C, device=DEV))
        y.fill_(float("nan"))
        assert lib.sn_plan_exec_launch(ex, plan, bases, 2, stream, None) == 0
        got = y.clone()
        y.fill_(float("nan"))
        assert lib.sn_plan_run(plan, bases, 2, stream, None) == 0
        assert torch.equal(got, y) and torch.equal(got, want())
    g = torch.cuda.CUDAGraph()                                     # the caller captures: the launch walks the list into ITS graph
    with torch.cuda.graph(g):
        assert lib.sn_plan_exec_launch(ex, plan, bases, 2, torch.cuda.current_stream().cuda_stream, None) == 0
    for seed in range(3):
        x.copy_(torch.randn(rows, C, device=DEV))
        y.fill_(float("nan"))
        g.replay()
        assert torch.equal(y, want())
    del g
    assert lib.sn_timing_enable(1) == 0
    try:
        ex2 = ctypes.c_void_p()
        assert lib.sn_plan_instantiate(plan, bases, 2, ctypes.byref(ex2), None) == -7 and not ex2      # SN_E_UNSUPPORTED
        y.fill_(float("nan"))
        assert lib.sn_plan_exec_launch(ex, plan, bases, 2, stream, None) == 0                           # (walks the list)
        assert torch.equal(y, want())
    finally:
        lib.sn_timing_enable(0)
        meta = (ctypes.c_int64 * 64)()
        ms = (ctypes.c_double * 8)()
        wr = ctypes.c_int64()
        lib.sn_timing_drain(ms, meta, 8, ctypes.byref(wr))
    torch.cuda.synchronize()
    assert lib.sn_plan_exec_destroy(ex) == 0 and lib.sn_plan_destroy(plan) == 0
