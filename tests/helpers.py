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


# ---- deterministic, library-independent parameter fill (golden fixtures store no weights) ------------------
def _splitmix_uniform(n, seed):
    """n doubles in [0,1) from splitmix64 on (index, seed): exact integer arithmetic, identical everywhere."""
    with np.errstate(over="ignore"):
        z = np.arange(n, dtype=np.uint64) + np.uint64((seed * 0x9E3779B97F4A7C15) % (1 << 64))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def det_tensor(shape, seed, scale=1.0):
    n = int(np.prod(shape)) if len(shape) else 1
    return ((_splitmix_uniform(n, seed) - 0.5) * 2 * scale).astype(np.float32).reshape(shape)


def deterministic_init(module, seed=0):
    """Fill every parameter/buffer of a torch module from splitmix64, keyed by its position in state_dict()."""
    import torch

    sd = module.state_dict()
    new = {}
    for i, (k, v) in enumerate(sd.items()):
        if k.endswith("num_batches_tracked"):
            new[k] = torch.zeros_like(v)
        elif k.endswith("running_var"):
            new[k] = torch.from_numpy(1.0 + 0.5 * det_tensor(tuple(v.shape), seed * 1000 + i, 1.0))
        elif k.endswith("bn.weight"):
            new[k] = torch.from_numpy(1.0 + 0.3 * det_tensor(tuple(v.shape), seed * 1000 + i, 1.0))
        else:
            fan_in = v.shape[1] if v.dim() == 2 else 8
            new[k] = torch.from_numpy(det_tensor(tuple(v.shape), seed * 1000 + i, 1.0 / np.sqrt(fan_in)))
    module.load_state_dict(new)
    return module


def grad_signature(module):
    """Per-parameter [L2 norm, sum, first 4 entries] of .grad — a compact stand-in for the full gradient."""
    out = {}
    for k, p in module.named_parameters():
        g = p.grad.detach().double().cpu().reshape(-1)
        head = np.zeros(4)
        head[: min(4, g.numel())] = g[:4].numpy()
        out[k] = np.concatenate([[float(g.norm()), float(g.sum())], head])
    return out


def sigs_close(sigs, ref_of, tol=1e-4):
    """Gradient signatures agree when every entry is within tol * (L2 norm of that gradient) + 1e-5 * (largest
    gradient norm in the model).  The floor matters for parameters whose exact gradient is zero (a Linear bias that
    feeds a BatchNorm): their computed gradient is pure round-off.  Returns the list of offending keys."""
    top = max(abs(float(ref_of(k)[0])) for k in sigs)
    bad = []
    for k, s in sigs.items():
        r = np.asarray(ref_of(k))
        if not (np.abs(np.asarray(s) - r) <= tol * abs(r[0]) + 1e-5 * top).all():
            bad.append((k, s, r))
    return bad
