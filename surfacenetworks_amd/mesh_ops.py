"""Sparse-direct construction of the Surface-Network operators from a triangle mesh (V, F).

Host-side (numpy/scipy, fp64 like the reference, cast to fp32 by the caller) restatement of the
reference's *dense* builders, which allocate O(F*V) memory and cannot reach the 5k-20k vertex
configurations:

    edge_lengths        <- mesh.dist                 src/utils/mesh.py:17-26
    heron_areas         <- mesh.area                 src/utils/mesh.py:67-80
    cotangent_weights   <- mesh.cotangent_weights    src/utils/mesh.py:102-112
    laplacian           <- graph.laplacian(normalized=False) then A^-1 * L
                                                     src/utils/graph.py:40-49, src/mesh_mnist/add_laplacian.py:47-48
    dirac               <- mesh.dirac                src/utils/mesh.py:35-64  (Q: mesh.py:28-33)

Everything is vectorised over faces; no dense (V,V) or (4F,4V) array is ever formed.  The results
are pinned against the imported reference on the golden fixtures (tests/golden/make_golden.py).

Also here: the synthetic mesh generators SURVEY.md §8(d) prescribes for the benchmark configs
(grid-cloth, torus-grid, seeded Delaunay) — there is no dataset in the reference tree.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

__all__ = [
    "edge_lengths", "heron_areas", "cotangent_weights", "laplacian", "dirac", "mesh_operators",
    "grid_cloth", "torus_grid", "delaunay_disc", "read_ply_ascii",
    "locality_order", "edge_span", "MeshOrder", "permute_operator",
]

_PERMS = np.array([[0, 1, 2], [0, 2, 1], [1, 0, 2], [1, 2, 0], [2, 0, 1], [2, 1, 0]])  # itertools.permutations order


def edge_lengths(V: np.ndarray, F: np.ndarray) -> np.ndarray:
    """(F,3) lengths l[f] = (|v0 v1|, |v1 v2|, |v2 v0|)  — the entries mesh.dist stores at (i,j)."""
    V = np.asarray(V, dtype=np.float64)
    i, j, k = F[:, 0], F[:, 1], F[:, 2]

    def d(a, b):
        return np.sqrt(((V[a] - V[b]) ** 2).sum(axis=1))

    return np.stack([d(i, j), d(j, k), d(k, i)], axis=1)


def heron_areas(l: np.ndarray) -> np.ndarray:
    """Heron's formula with the reference's 1e-6 floor for degenerate faces (mesh.py:75-78)."""
    lij, ljk, lki = l[:, 0], l[:, 1], l[:, 2]
    s = (lij + ljk + lki) / 2
    q = s * (s - lij) * (s - ljk) * (s - lki)
    out = np.full(l.shape[0], 1e-6)
    pos = q > 0
    out[pos] = np.sqrt(q[pos])
    return out


def cotangent_weights(F: np.ndarray, a: np.ndarray, l: np.ndarray, num_vertices: int):
    """W (symmetric cotangent weights, CSR) and A^-1 = diag(1/(A+1e-9)) as mesh.cotangent_weights.

    For every face and every ordered pair (i,j) of its vertices with k the third one:
        W[i,j] += (-l_ij^2 + l_jk^2 + l_ki^2) / (8 a_f + 1e-6);   A[i] += a_f/3/4  (once per permutation)
    """
    nF = F.shape[0]
    l2 = l ** 2                      # squares of (01, 12, 20)
    # squared length of the edge between local corners p,q
    e2 = np.empty((nF, 3, 3))
    e2[:, 0, 1] = e2[:, 1, 0] = l2[:, 0]
    e2[:, 1, 2] = e2[:, 2, 1] = l2[:, 1]
    e2[:, 2, 0] = e2[:, 0, 2] = l2[:, 2]
    rows, cols, vals = [], [], []
    denom = 8 * a + 1e-6
    for p, q, r in _PERMS:
        rows.append(F[:, p])
        cols.append(F[:, q])
        vals.append((-e2[:, p, q] + e2[:, q, r] + e2[:, r, p]) / denom)
    rows = np.stack(rows, axis=1).ravel()    # face-major, permutation-minor == reference loop order
    cols = np.stack(cols, axis=1).ravel()
    vals = np.stack(vals, axis=1).ravel()
    W = sp.coo_matrix((vals, (rows, cols)), shape=(num_vertices, num_vertices)).tocsr()
    W.eliminate_zeros()              # csr_matrix(dense) in the reference drops exact zeros
    A = np.zeros(num_vertices)
    np.add.at(A, rows, np.repeat(a / 3 / 4, 6))
    return W, sp.diags(1 / (A + 1e-9), 0)


def laplacian(V: np.ndarray, F: np.ndarray) -> sp.csr_matrix:
    """Mass-normalised cotangent Laplacian  L = A^-1 (D - W),  D = diag(column sums of W)."""
    V = np.asarray(V, dtype=np.float64)
    l = edge_lengths(V, F)
    a = heron_areas(l)
    W, Ainv = cotangent_weights(F, a, l, V.shape[0])
    d = np.asarray(W.sum(axis=0)).squeeze()
    L = sp.diags(d, 0) - W
    return (Ainv * L).tocsr()


def dirac(V: np.ndarray, F: np.ndarray):
    """Quaternionic Dirac operator D (4F x 4V) and its adjoint DA (4V x 4F) as mesh.dirac.

    Block (face i, corner vertex j), e = V[next] - V[next-next], a = 0:
        D [4i:4i+4, 4j:4j+4] = -Q(0,e) / (2 Af_i),     Q = [[a,-b,-c,-d],[b,a,-d,c],[c,d,a,-b],[d,-c,b,a]]
        DA[4j:4j+4, 4i:4i+4] = D_block^T * Af_i / Av_j,  Av_j = sum over incident faces of Af/3
    """
    V = np.asarray(V, dtype=np.float64)
    nV, nF = V.shape[0], F.shape[0]
    Af = heron_areas(edge_lengths(V, F))
    Av = np.zeros(nV)
    np.add.at(Av, F.ravel(), np.repeat(Af / 3, 3))       # face-major, corner-minor like the reference loop
    # (row r, col c, sign, component of e) of the 12 non-zeros of Q(0,e) with e = (0,b,c,d) -> index 0,1,2
    q_pat = [(0, 1, -1, 0), (0, 2, -1, 1), (0, 3, -1, 2),
             (1, 0, +1, 0), (1, 2, -1, 2), (1, 3, +1, 1),
             (2, 0, +1, 1), (2, 1, +1, 2), (2, 3, -1, 0),
             (3, 0, +1, 2), (3, 1, -1, 1), (3, 2, +1, 0)]
    fi = np.arange(nF)
    d_rows, d_cols, d_vals, da_vals = [], [], [], []
    for corner in range(3):
        j = F[:, corner]
        e = V[F[:, (corner + 1) % 3]] - V[F[:, (corner + 2) % 3]]      # (F,3)
        scale = 2 * Af
        for r, c, sgn, comp in q_pat:
            m = -(sgn * e[:, comp]) / scale                # entry of  -Q/(2Af)
            d_rows.append(4 * fi + r)
            d_cols.append(4 * j + c)
            d_vals.append(m)
            da_vals.append(m * Af / Av[j])                 # mat.T * Af / Av lands at (4j+c, 4i+r)
    d_rows = np.concatenate(d_rows)
    d_cols = np.concatenate(d_cols)
    D = sp.coo_matrix((np.concatenate(d_vals), (d_rows, d_cols)), shape=(4 * nF, 4 * nV)).tocsr()
    DA = sp.coo_matrix((np.concatenate(da_vals), (d_cols, d_rows)), shape=(4 * nV, 4 * nF)).tocsr()
    D.eliminate_zeros()
    DA.eliminate_zeros()
    D.sort_indices()
    DA.sort_indices()
    return D, DA


def mesh_operators(V: np.ndarray, F: np.ndarray, dtype=np.float32):
    """{'L','Di','DiA'} in CSR with `dtype` values — the per-sample record the reference stores
    (src/mesh_mnist/add_laplacian.py:63-71, src/as_rigid_as_possible/add_laplacian.py:61-65)."""
    L = laplacian(V, F)
    Di, DiA = dirac(V, F)
    return {"L": L.astype(dtype), "Di": Di.astype(dtype), "DiA": DiA.astype(dtype)}


# --------------------------------------------------------------------------------------------------
# synthetic meshes
# --------------------------------------------------------------------------------------------------
def _grid_faces(n: int, m: int, wrap: bool) -> np.ndarray:
    if wrap:
        ii, jj = np.meshgrid(np.arange(n), np.arange(m), indexing="ij")
        ni, nj = (ii + 1) % n, (jj + 1) % m
    else:
        ii, jj = np.meshgrid(np.arange(n - 1), np.arange(m - 1), indexing="ij")
        ni, nj = ii + 1, jj + 1
    v00 = (ii * m + jj).ravel()
    v01 = (ii * m + nj).ravel()
    v10 = (ni * m + jj).ravel()
    v11 = (ni * m + nj).ravel()
    F = np.empty((2 * v00.size, 3), dtype=np.int64)
    F[0::2] = np.stack([v00, v01, v11], axis=1)
    F[1::2] = np.stack([v00, v11, v10], axis=1)
    return F


def grid_cloth(n: int, m: int, rng: np.random.Generator, jitter: float = 0.25, permute: bool = False):
    """Open n x m jittered grid with a smooth height field: V = n*m, F = 2(n-1)(m-1). Row-major vertex order
    (or a seeded random permutation of it, to expose gather locality)."""
    ii, jj = np.meshgrid(np.arange(n, dtype=np.float64), np.arange(m, dtype=np.float64), indexing="ij")
    x = ii + jitter * (rng.random((n, m)) - 0.5)
    y = jj + jitter * (rng.random((n, m)) - 0.5)
    ph = rng.random(4) * 2 * np.pi
    z = 0.15 * n * (np.sin(2 * np.pi * ii / n + ph[0]) * np.cos(2 * np.pi * jj / m + ph[1])
                    + 0.5 * np.sin(4 * np.pi * ii / n + ph[2]) * np.sin(2 * np.pi * jj / m + ph[3]))
    V = np.stack([x.ravel(), y.ravel(), z.ravel()], axis=1) / max(n, m)
    F = _grid_faces(n, m, wrap=False)
    return _maybe_permute(V, F, rng, permute)


def torus_grid(n: int, m: int, rng: np.random.Generator, jitter: float = 0.1, permute: bool = False):
    """Closed n x m torus: V = n*m, F = 2*n*m (every vertex has degree 6)."""
    ii, jj = np.meshgrid(np.arange(n, dtype=np.float64), np.arange(m, dtype=np.float64), indexing="ij")
    u = 2 * np.pi * (ii + jitter * (rng.random((n, m)) - 0.5)) / n
    w = 2 * np.pi * (jj + jitter * (rng.random((n, m)) - 0.5)) / m
    R, r = 1.0, 0.35
    V = np.stack([((R + r * np.cos(w)) * np.cos(u)).ravel(), ((R + r * np.cos(w)) * np.sin(u)).ravel(),
                  (r * np.sin(w)).ravel()], axis=1)
    F = _grid_faces(n, m, wrap=True)
    return _maybe_permute(V, F, rng, permute)


def delaunay_disc(num_vertices: int, rng: np.random.Generator):
    """Mesh-MNIST-like open mesh: seeded random points on the 27x27 image plane, Delaunay-triangulated,
    intensity-like z, then the reference scaling V/27 - (0.5,0.5,0) (src/mesh_mnist/add_laplacian.py:40-41)."""
    from scipy.spatial import Delaunay

    pts = rng.random((num_vertices, 2)) * 27.0
    tri = Delaunay(pts)
    z = 0.5 + 0.5 * np.sin(pts[:, 0] / 4.0) * np.cos(pts[:, 1] / 5.0)
    V = np.concatenate([pts, z[:, None]], axis=1)
    V = V / 27 - np.array([0.5, 0.5, 0.0])
    return V, tri.simplices.astype(np.int64)


def _maybe_permute(V, F, rng, permute):
    """permute=True / "vertices": a seeded random renumbering of the vertices (SURVEY.md §8d's "random-permuted variant"; the
    face list keeps the generator's order); "both": the face list is shuffled as well (what a scanned mesh looks like)."""
    if not permute:
        return V, F
    perm = rng.permutation(V.shape[0])       # new index of old vertex i is inv[i]
    inv = np.empty_like(perm)
    inv[perm] = np.arange(perm.size)
    V, F = V[perm], inv[F]
    if permute == "both":
        F = F[rng.permutation(F.shape[0])]
    return V, F


# --------------------------------------------------------------------------------------------------
# locality order: datasets number their vertices arbitrarily (src/utils/mesh.py:35-64 builds every operator in the
# dataset's own order; FAUST scans are loaded as stored, src/dense_correspondence/main.py:66-102)
# --------------------------------------------------------------------------------------------------
def edge_span(F: np.ndarray, rank=None):
    """(mean, max) of |rank_i - rank_j| over the mesh edges: how far apart in memory the dense rows are that one operator row
    touches.  A row-major n x m grid has (2(m+1)/3, m+1); a random numbering about (V/3, V)."""
    F = np.asarray(F)
    r = F if rank is None else np.asarray(rank)[F]
    e = np.abs(np.concatenate([r[:, 0] - r[:, 1], r[:, 1] - r[:, 2], r[:, 2] - r[:, 0]]))
    return (float(e.mean()), int(e.max())) if e.size else (0.0, 0)


def locality_order(F: np.ndarray, num_vertices: int, num_faces_rows=None):
    """(vorder, forder): a numbering of the vertices and faces of a mesh under which every operator of the path is banded.
    vorder[k] = the dataset's index of the vertex stored at position k (reverse Cuthill-McKee on the vertex adjacency: the
    band of a 2-D mesh becomes ~ the width of its breadth-first level sets, <= the row-major band of a grid); forder[k]
    likewise for faces, sorted by the ranks of their corners (smallest first), so that face row ~ 2 x vertex row as in a
    generated grid.  One-time host work per mesh (milliseconds at 20 000 vertices)."""
    from scipy.sparse.csgraph import reverse_cuthill_mckee

    F = np.asarray(F, dtype=np.int64)
    nV = int(num_vertices)
    i = np.concatenate([F[:, 0], F[:, 1], F[:, 2]])
    j = np.concatenate([F[:, 1], F[:, 2], F[:, 0]])
    A = sp.coo_matrix((np.ones(i.size, np.int8), (i, j)), shape=(nV, nV)).tocsr()
    A = (A + A.T).tocsr()
    vorder = np.asarray(reverse_cuthill_mckee(A, symmetric_mode=True), dtype=np.int64)
    rank = np.empty(nV, np.int64)
    rank[vorder] = np.arange(nV)
    r = np.sort(rank[F], axis=1)
    forder = np.lexsort((r[:, 2], r[:, 1], r[:, 0])).astype(np.int64)
    return vorder, forder


class MeshOrder:
    """The stored numbering of one mesh: `vorder` / `forder` (stored position -> dataset index) and the inverse `vrank`
    (dataset index -> stored position).  `identity` when the dataset's own numbering was kept."""

    __slots__ = ("vorder", "forder", "vrank", "identity", "span_before", "span_after")

    def __init__(self, vorder, forder, identity=False, span_before=None, span_after=None):
        self.vorder, self.forder = np.asarray(vorder, np.int64), np.asarray(forder, np.int64)
        self.vrank = np.empty_like(self.vorder)
        self.vrank[self.vorder] = np.arange(self.vorder.size)
        self.identity = bool(identity)
        self.span_before, self.span_after = span_before, span_after

    @classmethod
    def of_mesh(cls, F, num_vertices: int, mode="auto", gain: float = 2.0) -> "MeshOrder":
        """mode False / "off": keep the dataset's numbering; True / "rcm": always renumber; "auto" (default): renumber when
        the mean edge span shrinks by more than `gain` x (a generated grid stays as it is, a scan or a permuted grid does
        not)."""
        F = np.asarray(F)
        nV, nF = int(num_vertices), int(F.shape[0])
        if mode in (False, None, "off"):
            return cls(np.arange(nV), np.arange(nF), identity=True)
        vorder, forder = locality_order(F, nV)
        rank = np.empty(nV, np.int64)
        rank[vorder] = np.arange(nV)
        before, after = edge_span(F)[0], edge_span(F, rank)[0]
        if mode == "auto" and before <= gain * after:
            return cls(np.arange(nV), np.arange(nF), identity=True, span_before=before, span_after=before)
        return cls(vorder, forder, span_before=before, span_after=after)

    def mesh(self, V, F):
        """(V, F) in the stored numbering; V may carry leading axes (frames): (..., nV, 3)."""
        if self.identity:
            return V, F
        return np.asarray(V)[..., self.vorder, :], self.vrank[np.asarray(F)[self.forder]]

    def vertex_rows(self, x):
        """Per-vertex data (nV, ...) from the dataset's numbering into the stored one."""
        return x if self.identity else np.asarray(x)[self.vorder]


def permute_operator(A, row_order, col_order, group: int = 1):
    """P_r A P_c^T for a stored operator (the reference's dataset files hold L, Di, DiA per frame): row k of the result is row
    row_order[k] of A, likewise columns; group = 4 for the quaternion operators (blocks of 4 rows / columns move together)."""
    def expand(o):
        o = np.asarray(o, np.int64)
        return o if group == 1 else (group * o[:, None] + np.arange(group)[None, :]).ravel()
    A = A.tocsr()
    out = A[expand(row_order)][:, expand(col_order)].tocsr()
    out.sort_indices()
    return out


def read_ply_ascii(path: str):
    """Minimal ASCII PLY reader (vertex xyz + triangular faces) for meshes/cube.ply-style files."""
    with open(path) as fh:
        lines = [ln.strip() for ln in fh]
    nv = nf = 0
    k = 0
    while lines[k] != "end_header":
        tok = lines[k].split()
        if tok[:2] == ["element", "vertex"]:
            nv = int(tok[2])
        if tok[:2] == ["element", "face"]:
            nf = int(tok[2])
        k += 1
    k += 1
    V = np.array([[float(t) for t in lines[k + i].split()[:3]] for i in range(nv)])
    F = np.array([[int(t) for t in lines[k + nv + i].split()[1:4]] for i in range(nf)], dtype=np.int64)
    return V, F
