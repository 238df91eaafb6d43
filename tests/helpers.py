"""Shared test helpers: seeded meshes/operators and comparison utilities."""
import numpy as np
import scipy.sparse as sp

from surfacenetworks_amd import mesh_ops


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def mesh_fixture(kind, seed=0):
    rng = np.random.default_rng(seed)
    if kind == "cloth":
        V, F = mesh_ops.grid_cloth(13, 9, rng)
    elif kind == "cloth_perm":
        V, F = mesh_ops.grid_cloth(11, 10, rng, permute=True)
    elif kind == "torus":
        V, F = mesh_ops.torus_grid(9, 12, rng)
    elif kind == "delaunay":
        V, F = mesh_ops.delaunay_disc(150, rng)
    else:
        raise KeyError(kind)
    return V, F, mesh_ops.mesh_operators(V, F)


def random_csr(M, K, density, seed, empty_rows=()):
    A = sp.random(M, K, density=density, format="lil", dtype=np.float32, random_state=seed)
    for r in empty_rows:
        A.rows[r] = []
        A.data[r] = []
    A = A.tocsr()
    A.sort_indices()
    return A
