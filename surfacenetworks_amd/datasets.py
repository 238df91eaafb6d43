"""Readers for the reference's on-disk dataset formats -> device-resident pools.

The reference stores its preprocessed data as pickled numpy object arrays:

  ARAP        as_rigid_as_possible/data_plus/<seq>.npy  : list of 50 frame dicts {'V' (nV,3) f32, 'F' (nF,3) i32} with
              {'L','Di','DiA'} scipy CSR float32 on the first 10 frames
              (written by src/as_rigid_as_possible/add_laplacian.py:61-70, read by main.py:58-94)
  Mesh-MNIST  mesh_mnist/data/{train,test}_plus.np       : list of dicts {'V','F','L','flat_L','Di','DiA','flat_Di',
              'flat_DiA','label'}  (src/mesh_mnist/add_laplacian.py:63-71, read by main.py:54-72)
  FAUST       *.npz with V, F, L, D, DA (0-d object arrays holding scipy matrices), label, label_inv, dist_mat
              (read by src/dense_correspondence/main.py:65-104)

Loading them needs `allow_pickle=True` (they ARE pickles); only load files you trust.  The writers below produce the
same layouts from synthetic meshes so that the readers are testable without the (undownloadable) datasets.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np
import torch

from . import arap as _arap
from . import mesh_ops
from .operators import OperatorPool

__all__ = ["load_arap_sequence", "arap_from_files", "load_mesh_mnist", "mnist_from_samples", "load_faust_frame", "faust_from_files",
           "write_arap_sequence", "write_mesh_mnist", "write_faust_frame"]


# ---- ARAP ---------------------------------------------------------------------------------------------------
def load_arap_sequence(path: str) -> List[Dict]:
    """One <seq>.npy as a list of frame dicts (numpy / scipy objects, untouched)."""
    seq = np.load(path, encoding="latin1", allow_pickle=True)          # as main.py:59
    return list(seq)


def arap_from_files(paths: Sequence[str], device="cuda", model="dir", reorder="auto") -> "_arap.ClothSequences":
    """Build the resident ARAP dataset (frame coordinates + operator pools) from reference sequence files.
    Returns an object with the ClothSequences interface (sample_batch, ...).
    reorder ("auto" / True / False): the sequences are STORED in a locality numbering of their vertices and faces
    (mesh_ops.MeshOrder; the files keep whatever order the scanner / simulator wrote): coordinates are permuted once, the
    stored operators become P_r A P_c^T; `ds.to_dataset_order(outputs, seq_ids)` maps per-vertex results back."""
    seqs = [load_arap_sequence(p) for p in paths]
    ds = _arap.ClothSequences.__new__(_arap.ClothSequences)
    ds.device = torch.device(device)
    ds.kind, ds.operators = model, "pool"
    ds.n = len(seqs)
    ds.frames = min(len(s) for s in seqs)
    ds.op_frames = min(sum(1 for fr in s if fr.get("Di" if model == "dir" else "L") is not None) for s in seqs)
    ds.num_vertices = np.array([s[0]["V"].shape[0] for s in seqs])
    ds.num_faces = np.array([s[0]["F"].shape[0] for s in seqs])
    ds.orders = [mesh_ops.MeshOrder.of_mesh(np.asarray(s[0]["F"]), s[0]["V"].shape[0], reorder) for s in seqs]
    vmax = int(ds.num_vertices.max())
    xyz = np.zeros((ds.n, ds.frames, vmax, 3), np.float32)
    for i, s in enumerate(seqs):
        for t in range(ds.frames):
            xyz[i, t, : s[t]["V"].shape[0]] = ds.orders[i].vertex_rows(np.asarray(s[t]["V"], dtype=np.float32))
    ds.xyz = torch.from_numpy(xyz).to(ds.device)
    ds.vcount = torch.from_numpy(ds.num_vertices).to(ds.device)

    def pick(key, rows, cols, group):
        out = []
        for s, o in zip(seqs, ds.orders):
            for t in range(ds.op_frames):
                A = s[t][key].astype(np.float32)
                out.append(A if o.identity else mesh_ops.permute_operator(A, getattr(o, rows), getattr(o, cols), group))
        return out
    if model == "dir":
        ds.pool_Di = OperatorPool(pick("Di", "forder", "vorder", 4), ds.device, want_bsr4=True)
        ds.pool_DiA = OperatorPool(pick("DiA", "vorder", "forder", 4), ds.device, want_bsr4=True)
    else:
        ds.pool_L = OperatorPool(pick("L", "vorder", "vorder", 1), ds.device)
    return ds


def write_arap_sequence(path: str, Vt: np.ndarray, F: np.ndarray, op_frames: int = 10) -> None:
    """Write a sequence in the reference layout (add_laplacian.py:45-70): operators on the first `op_frames` frames."""
    frames = []
    for t in range(Vt.shape[0]):
        fr = {"V": Vt[t].astype("float32"), "F": F.astype("int32")}
        if t < op_frames:
            ops = mesh_ops.mesh_operators(Vt[t].astype(np.float64), F)
            fr.update({"L": ops["L"], "Di": ops["Di"], "DiA": ops["DiA"]})
        frames.append(fr)
    arr = np.empty(len(frames), dtype=object)
    arr[:] = frames
    with open(path, "wb") as fh:
        np.save(fh, arr, allow_pickle=True)


# ---- Mesh-MNIST --------------------------------------------------------------------------------------------------
def load_mesh_mnist(path: str) -> List[Dict]:
    with open(path, "rb") as fh:                                        # as main.py:54
        return list(np.load(fh, encoding="latin1", allow_pickle=True))


def mnist_from_samples(samples: Sequence[Dict], device="cuda", model="dir", reorder="auto"):
    """Resident Mesh-MNIST dataset (MeshDigits interface) from reference sample dicts; `reorder`: stored in a locality numbering
    (mesh_ops.MeshOrder — the file's operators become P_r A P_c^T; the model's output is per mesh, nothing maps back)."""
    from . import mesh_mnist as mm

    orders = [mesh_ops.MeshOrder.of_mesh(np.asarray(s["F"]), s["V"].shape[0], reorder) for s in samples]

    def op(i, key, rows, cols, group):
        A, o = samples[i][key].astype(np.float32), orders[i]
        return A if o.identity else mesh_ops.permute_operator(A, getattr(o, rows), getattr(o, cols), group)

    ds = mm.MeshDigits.__new__(mm.MeshDigits)
    ds.device, ds.kind, ds.n = torch.device(device), model, len(samples)
    ds.nv = np.array([s["V"].shape[0] for s in samples])
    ds.nf = np.array([s["F"].shape[0] for s in samples])
    xyz = np.zeros((ds.n, int(ds.nv.max()), 3), np.float32)
    for i, s in enumerate(samples):
        xyz[i, : ds.nv[i]] = orders[i].vertex_rows(np.asarray(s["V"], dtype=np.float32))
    ds.xyz = torch.from_numpy(xyz).to(ds.device)
    ds.vcount = torch.from_numpy(ds.nv).to(ds.device)
    ds.labels = torch.tensor([int(s["label"]) for s in samples], device=ds.device)
    if model == "dir":
        ds.pool_Di = OperatorPool([op(i, "Di", "forder", "vorder", 4) for i in range(ds.n)], ds.device, want_bsr4=True)
        ds.pool_DiA = OperatorPool([op(i, "DiA", "vorder", "forder", 4) for i in range(ds.n)], ds.device, want_bsr4=True)
    else:
        ds.pool_L = OperatorPool([op(i, "L", "vorder", "vorder", 1) for i in range(ds.n)], ds.device)
    ds.orders = orders
    ds.run_nv = ds.run_nf = 0
    return ds


def write_mesh_mnist(path: str, meshes: Sequence, labels: Sequence[int]) -> None:
    """meshes: iterable of (V, F).  Layout of add_laplacian.py:63-71 (flat_* variants included)."""
    out = []
    for (V, F), lab in zip(meshes, labels):
        ops = mesh_ops.mesh_operators(V, F)
        Vf = V.copy()
        Vf[:, 2] = 0
        flat = mesh_ops.mesh_operators(Vf, F)
        out.append({"V": V.astype("float32"), "F": F.astype("int32"), "L": ops["L"], "flat_L": flat["L"], "Di": ops["Di"],
                    "DiA": ops["DiA"], "flat_Di": flat["Di"], "flat_DiA": flat["DiA"], "label": int(lab)})
    arr = np.empty(len(out), dtype=object)
    arr[:] = out
    with open(path, "wb") as fh:
        np.save(fh, arr, allow_pickle=True)


# ---- FAUST ----------------------------------------------------------------------------------------------------------
def load_faust_frame(path: str, device="cuda") -> Dict:
    """One FAUST .npz as the frame dict of dense_correspondence/main.py:65-102 (tensors on `device`, scipy operators)."""
    with np.load(path, allow_pickle=True) as z:
        fr = {
            "V": torch.from_numpy(z["V"].astype("f")).to(device),
            "F": torch.from_numpy(z["F"].astype(np.int64)).to(device),
            "L": z["L"].item().astype("f").tocsr() if "L" in z else None,
            "Di": z["D"].item().astype("f").tocsr() if "D" in z else None,
            "DiA": z["DA"].item().astype("f").tocsr() if "DA" in z else None,
            "label": torch.from_numpy(z["label"]).to(device),
            "label_inv": torch.from_numpy(z["label_inv"]).to(device),
            "G": torch.from_numpy(z["dist_mat"].astype("f")).to(device),
        }
    return fr


def faust_from_files(paths: Sequence[str], device="cuda", model="lap", pad_to=None, reorder="auto"):
    """Resident FAUST dataset (dense_correspondence.FaustFrames) from reference .npz frames; `reorder` as FaustFrames."""
    from . import dense_correspondence as dc

    return dc.FaustFrames([load_faust_frame(p, device) for p in paths], model=model, pad_to=pad_to, device=device, reorder=reorder)


def write_faust_frame(path: str, V: np.ndarray, F: np.ndarray, label: np.ndarray, dist_mat: np.ndarray) -> None:
    ops = mesh_ops.mesh_operators(V, F)
    wrap = lambda m: np.array(m, dtype=object)
    np.savez(path, V=V, F=F, L=wrap(ops["L"]), D=wrap(ops["Di"]), DA=wrap(ops["DiA"]), label=label,
             label_inv=np.argsort(label), dist_mat=dist_mat)
