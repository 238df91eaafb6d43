"""Launch plans, host side (no GPU): the entry table of csrc/sn_plan.hip is what tools/gen_plan_table.py writes from the ctypes
signatures; the C plan object checks argument kinds against the entry points' parameter types; the dry run of the reference
blocks on CPU tensors records launches only (no torch op that computes), resolves every pointer into an arena or an operand,
and leaves the operands' attributes and the table of gradient bounds as it found them."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_table_is_what_the_generator_writes():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_plan_table

    with open(gen_plan_table.OUT) as fh:
        assert fh.read() == gen_plan_table.text()


def test_entry_signatures_match_the_ctypes_table():
    from surfacenetworks_amd import _lib

    lib = _lib.load()
    n = int(lib.sn_plan_entry_count())
    assert n >= 80
    for i in range(n):
        name = lib.sn_plan_entry_name(i).decode()
        args = _lib.SIGNATURES[name][1]
        want = "".join("p" if a is C.c_void_p else "d" if a in (C.c_double, C.c_float) else "i" for a in args)
        assert lib.sn_plan_entry_signature(i).decode() == want, name
    assert int(lib.sn_plan_lookup(b"sn_timing_drain")) == -1 and int(lib.sn_plan_lookup(b"no_such_entry")) == -1


def test_add_call_rejects_wrong_kinds_and_counts():
    from surfacenetworks_amd import _lib

    lib = _lib.load()
    h = C.c_void_p()
    assert lib.sn_plan_create(C.byref(h)) == 0
    fn = int(lib.sn_plan_lookup(b"sn_elu_into_f32"))          # (src, lds, dst, ldd, rows, C, stream)
    n = 7
    arr = lambda ty, v: (ty * n)(*v)
    ok_kind = [2, 0, 2, 0, 0, 0, 4]
    slot = arr(C.c_int32, [2, 0, 0, 0, 0, 0, 0])
    ival = arr(C.c_int64, [0, 128, 256, 256, 10, 128, 0])
    dval = arr(C.c_double, [0.0] * n)
    assert lib.sn_plan_add_call(h, fn, n, arr(C.c_int32, ok_kind), slot, ival, dval) == 0
    assert lib.sn_plan_length(h) == 1
    bad = list(ok_kind)
    bad[1] = 2                                                   # a pointer where the leading dimension belongs
    assert lib.sn_plan_add_call(h, fn, n, arr(C.c_int32, bad), slot, ival, dval) == -7
    bad = list(ok_kind)
    bad[6] = 3                                                   # the stream must be the stream
    assert lib.sn_plan_add_call(h, fn, n, arr(C.c_int32, bad), slot, ival, dval) == -7
    assert lib.sn_plan_add_call(h, fn, n - 1, arr(C.c_int32, ok_kind), slot, ival, dval) == -2
    # running with too few slots / an empty slot is an argument error, not a launch
    failed = C.c_int32(-1)
    bases = (C.c_uint64 * 3)(0, 0, 0)
    assert lib.sn_plan_run(h, bases, 2, None, C.byref(failed)) == -2
    assert lib.sn_plan_run(h, bases, 3, None, C.byref(failed)) == -1 and failed.value == 0
    assert lib.sn_plan_destroy(h) == 0


def test_dry_run_of_the_blocks_records_launches_only(monkeypatch):
    from surfacenetworks_amd import blocks, functional as snF, kernels, mesh_ops, plans
    from surfacenetworks_amd import utils_pt as U
    from surfacenetworks_amd.operators import SparseOperator

    monkeypatch.setattr(plans, "_ALLOW_CPU", True)
    monkeypatch.setattr(plans.Plan, "run", lambda self, big, small, ext: None)        # (nothing can launch here)
    monkeypatch.setenv("SN_STRICT", "1")                                               # a refused plan raises
    plans.reset()
    snF.set_dirac_format("csr")                                                        # (the packed forms are built by device kernels)
    try:
        rng = np.random.default_rng(0)
        V, F = mesh_ops.grid_cloth(12, 9, rng)
        ops = mesh_ops.mesh_operators(V, F)
        nV, nF, Cc = V.shape[0], F.shape[0], 128

        def op_of(A):
            o = SparseOperator.from_scipy(A, "cpu")
            At = A.T.tocsr()
            At.sort_indices()
            o._t = SparseOperator.from_scipy(At, "cpu")
            return o

        Di, DiA, L = op_of(ops["Di"]), op_of(ops["DiA"]), op_of(ops["L"])
        for o in (L, L._t):
            o.format = "csr"
        b1, avg, b2, lap = U.DirResNet2(Cc), U.AvgResNet2(Cc), U.DirResNet2(Cc), U.LapResNet2(Cc)
        conv = U.GraphConv1x1(Cc, 120, batch_norm="pre")
        v = torch.randn(1, nV, Cc, requires_grad=True)
        mask = torch.ones(1, nV, 1)
        table_before = dict(kernels._absmax_table)
        for _ in range(2):                                                             # the recording call, then a replay
            v1, f1 = b1(Di, DiA, v, None, f_out_needed=False, num_faces=nF, avg_next=True)
            v2 = avg(None, mask, v1)
            v3, f3 = b2(Di, DiA, v2, f1, f_out_needed=True, num_faces=nF, avg_next=False)
            v4 = lap(L, mask, v3)
            y = U.elu_conv1x1(conv, v4)
            (y.sum() + f3.sum()).backward()
            kernels.clear_absmax()
        st = plans.stats()
        for site in ("dirac_fwd", "dirac_bwd", "avg_fwd", "avg_bwd", "propagate_fwd", "propagate_bwd", "elu_conv_fwd", "elu_conv_bwd"):
            assert st[site]["refused"] == 0 and st[site]["recorded"] >= 1 and st[site]["replayed"] >= 2, (site, st[site])
        assert st["dirac_fwd"]["recorded"] == 2                                          # zero-face block and the general one
        for site in blocks._SITES.values():
            for plan in site.plans.values():
                assert plan is not None and plan.launches >= 3
                assert plan.arena_bytes[0] % 256 == 0 and plan.arena_bytes[1] % 256 == 0
        assert kernels._absmax_table == {} and table_before == {}
        for bn in (b1.bn_fc0.bn, b1.bn_fc1.bn, avg.bn_fc0.bn, lap.bn_fc1.bn, conv.bn):
            assert "_sn_nbt" not in bn.running_mean.__dict__                          # the one-shot counter attribute never survives
    finally:
        snF.set_dirac_format("q3")
        plans.reset()


def test_sites_stop_recording_when_every_call_has_a_new_shape_and_keep_a_bounded_table(monkeypatch):
    """Random ragged batches: a new signature per call.  The site records a few, notices that nothing comes back, leaves the
    following calls to the eager path (plans.Site.lookup) — and a table never holds more than MAX_PLANS_PER_SITE plans."""
    from surfacenetworks_amd import blocks, plans
    from surfacenetworks_amd import utils_pt as U

    monkeypatch.setattr(plans, "_ALLOW_CPU", True)
    monkeypatch.setattr(plans.Plan, "run", lambda self, big, small, ext: None)
    monkeypatch.setattr(plans, "MAX_PLANS_PER_SITE", 6)
    plans.reset()
    site = blocks._SITES["avg_fwd"]
    fell_back = []
    real = blocks._avg_fwd

    def counting(*a):
        if plans._lib._recorder is None:
            fell_back.append(a[0].shape)
            raise _Eager()
        return real(*a)

    class _Eager(Exception):
        pass

    monkeypatch.setattr(blocks, "_avg_fwd", counting)
    avg = U.AvgResNet2(128)
    try:
        for i in range(40):
            nv = 40 + i                                           # a new shape every call
            x = torch.randn(1, nv, 128)
            try:
                avg(None, torch.ones(1, nv, 1), x)
            except _Eager:
                pass
        st = plans.stats()["avg_fwd"]
        assert st["recorded"] <= plans.THRASH_WINDOW and st["skipped"] >= 40 - plans.THRASH_WINDOW - 1, st
        assert len(fell_back) == st["skipped"]
        assert len(site.plans) <= 6
        # a shape that was recorded and is still in the table replays even during the cool-down
        nv = 40 + st["recorded"] - 1
        before = st["replayed"]
        avg(None, torch.ones(1, nv, 1), torch.randn(1, nv, 128))
        assert plans.stats()["avg_fwd"]["replayed"] == before + 1
    finally:
        plans.reset()


def test_a_dry_run_records_only_the_calls_of_its_own_thread(monkeypatch):
    """The recorder belongs to the thread that runs the dry run: a call of another thread during it (autograd's device threads, a
    loader thread) is a launch, not an entry of somebody else's plan — and a second dry run waits for the first."""
    import threading

    from surfacenetworks_amd import _lib, plans

    launched, order = [], []

    class _FakeLib:
        def __getattr__(self, name):
            return lambda *a: launched.append((name, threading.get_ident())) or 0

    monkeypatch.setattr(_lib, "load", lambda: _FakeLib())
    inside, release = threading.Event(), threading.Event()

    def impl():
        _lib.call("sn_mine", 1)                                  # recorded: this thread owns the recorder
        inside.set()
        assert release.wait(10)
        return None

    def other():
        assert inside.wait(10)
        assert _lib._recorder is not None and _lib.recorder() is None
        _lib.call("sn_theirs", 2)                                # launched
        second = threading.Thread(target=lambda: (plans._dry_run(lambda: order.append("second"), (), []), None))
        second.start()
        second.join(0.2)
        assert second.is_alive() and order == []                 # a second dry run waits for the first
        order.append("first ends")
        release.set()
        second.join(10)

    t = threading.Thread(target=other)
    t.start()
    rec, _ = plans._dry_run(impl, (), [])
    t.join(10)
    assert not t.is_alive()
    assert [n.name if hasattr(n, "name") else n[0] for n in rec.nodes] == ["sn_mine"] or len(rec.nodes) == 1
    assert [name for name, _ in launched] == ["sn_theirs"]
    assert order == ["first ends", "second"]
    assert _lib._recorder is None


def test_operands_that_overlap_in_memory_get_no_plan():
    """A recorded pointer is attributed to the first operand whose range holds it; operands that overlap during the dry run
    would be told apart wrongly on a call where they do not.  Same first byte among the positions whose sharing pattern the
    caller's key holds is the one exception."""
    from surfacenetworks_amd import plans

    buf = torch.zeros(4, 129)
    x, m = buf[:, :128], buf[:, 128:]                              # interleaved rows: disjoint elements, overlapping ranges
    other = torch.zeros(16)
    with pytest.raises(plans.PlanError, match="overlap"):
        plans._Layout(plans._Recorder([x, other, m]))
    with pytest.raises(plans.PlanError, match="overlap"):
        plans._Layout(plans._Recorder([other, other[4:]]))
    with pytest.raises(plans.PlanError, match="overlap"):
        plans._Layout(plans._Recorder([other, other]))             # the same memory twice, not keyed by the caller
    lay = plans._Layout(plans._Recorder([other, x, other]), alias_ok=(0, 2))
    assert lay.resolve(other.data_ptr() + 8) == (2 + 0, 8)
    assert plans._Layout(plans._Recorder([other[:8], other[8:], None, torch.zeros(0)])).resolve(other.data_ptr() + 32) == (3, 0)


def test_rent_or_buy_policy_of_plan_graphs(monkeypatch):
    """Plan._launch without a device: an address set walks the list until it has come back GRAPH_AFTER_SIGHTINGS times, then buys its
    graph and launches it; a capture of the caller's postpones the purchase; a refused instantiation or a failed graph launch sends
    the set back to the list for good; the table is bounded and evicted graphs are destroyed."""
    import collections
    import ctypes

    from surfacenetworks_amd import kernels, plans

    events = []

    class _Lib:
        fail_launch = False
        refuse = False

        def sn_plan_run(self, h, b, n, s, failed):
            events.append(("walk", bytes(b)))
            return 0

        def sn_plan_instantiate(self, h, b, n, out, failed):
            if self.refuse:
                return -7
            ctypes.cast(out, ctypes.POINTER(ctypes.c_void_p))[0] = 0x1000 + len(events)
            events.append(("buy", bytes(b)))
            return 0

        def sn_plan_exec_launch(self, x, h, b, n, s, failed):
            if self.fail_launch:
                return 1
            events.append(("graph", bytes(b)))
            return 0

        def sn_plan_exec_destroy(self, x):
            events.append(("destroy", x.value if hasattr(x, "value") else x))
            return 0

    capturing = [False]
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: capturing[0])
    monkeypatch.setattr(kernels, "_stream", lambda: 0)
    monkeypatch.setattr(plans, "_GRAPHS", True)
    monkeypatch.setattr(plans, "GRAPH_AFTER_SIGHTINGS", 3)
    monkeypatch.setattr(plans, "MAX_EXECS_PER_PLAN", 2)
    plans._graveyard.clear()
    lib = _Lib()
    plan = object.__new__(plans.Plan)
    plan._lib, plan.handle, plan.n_ext, plan.graphable = lib, ctypes.c_void_p(1), 1, True
    plan.execs = collections.OrderedDict()
    T = ctypes.c_uint64 * 3
    a, b, c = T(1, 2, 3), T(4, 5, 6), T(7, 8, 9)
    kinds = lambda: [e[0] for e in events]
    for _ in range(2):
        plan._launch(a)
    assert kinds() == ["walk", "walk"]
    capturing[0] = True
    plan._launch(a)                                                # third sighting, but the caller is capturing: not now
    assert kinds() == ["walk"] * 3
    capturing[0] = False
    plan._launch(a)
    plan._launch(a)
    assert kinds() == ["walk"] * 3 + ["buy", "graph", "graph"]
    lib.refuse = True
    for _ in range(5):
        plan._launch(b)
    assert kinds()[6:] == ["walk"] * 5 and plan.execs[bytes(b)] is plans._NEVER
    lib.refuse, lib.fail_launch = False, True
    plan._launch(a)                                                # the runtime will not launch the graph: the list, for good
    assert kinds()[11:] == ["walk"] and plan.execs[bytes(a)] is plans._NEVER and len(plans._graveyard) == 1
    lib.fail_launch = False
    plan._launch(a)
    assert kinds()[12:] == ["walk"]
    plan._launch(c)                                                # a third set: the least recently used one leaves the table
    assert len(plan.execs) == 2 and bytes(b) not in plan.execs
    plan.graphable = False
    plan._launch(c)
    assert plan.execs[bytes(c)] == 1                               # (large plans never count sightings)
    plan.handle = None
    plans._graveyard.clear()
