"""mesh_ops.MeshOrder — the locality numbering datasets are stored in (host logic; no GPU).

The reference builds every operator in the dataset's own vertex order (src/utils/mesh.py:35-64; FAUST scans are loaded as
stored, src/dense_correspondence/main.py:66-102); the product stores meshes in a banded numbering instead.  Checked here: the
numbering is a relabeling of the same mesh, operators of the relabeled mesh are P A P^T of the stored ones, "auto" leaves a
generated grid alone and renumbers a shuffled one."""
import numpy as np
import pytest

from surfacenetworks_amd import mesh_ops as mo


@pytest.mark.parametrize("kind", ["cloth_both", "cloth_vertices", "torus", "delaunay"])
def test_locality_order_is_a_relabeling_of_the_same_mesh(kind):
    rng = np.random.default_rng(3)
    if kind == "cloth_both":
        V, F = mo.grid_cloth(17, 12, rng, permute="both")
    elif kind == "cloth_vertices":
        V, F = mo.grid_cloth(17, 12, rng, permute=True)
    elif kind == "torus":
        V, F = mo.torus_grid(9, 14, rng, permute="both")
    else:
        V, F = mo.delaunay_disc(180, rng)
    nV, nF = V.shape[0], F.shape[0]
    o = mo.MeshOrder.of_mesh(F, nV, True)
    assert sorted(o.vorder.tolist()) == list(range(nV)) and sorted(o.forder.tolist()) == list(range(nF))
    assert np.array_equal(o.vrank[o.vorder], np.arange(nV))
    V2, F2 = o.mesh(V, F)
    # same geometry: stored vertex k is dataset vertex vorder[k]; stored face k is dataset face forder[k], corner for corner
    assert np.array_equal(V2, V[o.vorder])
    assert np.array_equal(o.vorder[F2], F[o.forder])
    # frames ride along: (T, nV, 3)
    Vt = np.stack([V, V + 1.0])
    assert np.array_equal(o.mesh(Vt, F)[0], Vt[:, o.vorder])
    assert mo.edge_span(F2)[0] < mo.edge_span(F)[0]
    # operators of the relabeled mesh are the relabeled operators (Di exactly: per-face arithmetic; DiA / L sum per vertex
    # over faces in another order: rounding)
    Di, DiA = mo.dirac(V, F)
    Di2, DiA2 = mo.dirac(V2, F2)
    assert abs(mo.permute_operator(Di, o.forder, o.vorder, 4) - Di2).max() == 0
    assert abs(mo.permute_operator(DiA, o.vorder, o.forder, 4) - DiA2).max() <= 1e-12 * abs(DiA2).max()
    L, L2 = mo.laplacian(V, F), mo.laplacian(V2, F2)
    assert abs(mo.permute_operator(L, o.vorder, o.vorder) - L2).max() <= 1e-10 * abs(L2).max()


def test_auto_keeps_a_generated_grid_and_renumbers_a_shuffled_one():
    rng = np.random.default_rng(0)
    V, F = mo.grid_cloth(71, 71, rng)
    keep = mo.MeshOrder.of_mesh(F, V.shape[0], "auto")
    assert keep.identity and np.array_equal(keep.vorder, np.arange(V.shape[0]))
    assert keep.mesh(V, F)[0] is V
    Vp, Fp = mo.grid_cloth(71, 71, rng, permute="both")
    o = mo.MeshOrder.of_mesh(Fp, Vp.shape[0], "auto")
    assert not o.identity
    rank = o.vrank
    # banded like the row-major grid: every edge within 72 positions (row-major: 72)
    assert mo.edge_span(Fp, rank)[1] <= 72 and mo.edge_span(Fp)[1] > 4000
    assert mo.MeshOrder.of_mesh(Fp, Vp.shape[0], False).identity
    # a closed mesh's wrap-around rows disappear too
    Vt, Ft = mo.torus_grid(65, 106, rng)
    ot = mo.MeshOrder.of_mesh(Ft, Vt.shape[0], True)
    assert mo.edge_span(Ft)[1] == 6889 and mo.edge_span(Ft, ot.vrank)[1] < 260


def test_faces_follow_their_vertices():
    rng = np.random.default_rng(2)
    V, F = mo.grid_cloth(30, 25, rng, permute="both")
    o = mo.MeshOrder.of_mesh(F, V.shape[0], True)
    _, F2 = o.mesh(V, F)
    lo = F2.min(axis=1)
    assert (np.diff(lo) >= 0).all()                        # sorted by the smallest corner rank
    # face k sits near 2 x (its vertices' rank): the Dirac blocks stay near the diagonal of the (F x V) block pattern
    assert np.abs(np.arange(F2.shape[0]) / 2.0 - F2.mean(axis=1)).max() < 3 * 30


def test_bench_config5_orders_are_what_they_say(monkeypatch):
    """bench.C5_ORDERS: the same mesh sizes in every variant; mean edge span large for the shuffled variants, back to (below) the
    grid's for the renumbered one; operators of every variant have the same entry counts."""
    import bench

    monkeypatch.setattr(bench, "C5_MESHES_PER_GPU", 6)
    monkeypatch.setattr(bench, "C5_VMIN", 300)
    monkeypatch.setattr(bench, "C5_VMAX", 900)
    res = {o: bench._c5_meshes(0, *pr) for o, pr in bench.C5_ORDERS.items()}
    sums = {o: (r[3], r[4]) for o, r in res.items()}
    assert len(set(sums.values())) == 1                                   # same sum V, sum F everywhere
    span = {o: r[5] for o, r in res.items()}
    assert span["permuted"] > 5 * span["grid"] and span["permuted_both"] > 5 * span["grid"]
    assert span["permuted_both+reorder"] <= span["grid"]
    nnz = {o: [m.nnz for m in r[0]] for o, r in res.items()}
    assert nnz["grid"] == nnz["permuted"] == nnz["permuted_both"] == nnz["permuted_both+reorder"]
