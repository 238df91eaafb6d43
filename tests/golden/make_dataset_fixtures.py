"""Generates the on-disk dataset fixtures tests/golden/data_*.{npy,np,npz} in the REFERENCE's file layouts, by running the
reference's own preprocessing functions in the build container, and checks the build's readers against the reference's.

    ARAP        src/as_rigid_as_possible/add_laplacian.py:39-70  `process(seqname)`  reads <seq>/00..49.obj with
                utils.mesh.load_obj and writes as_rigid_as_possible/data_plus/<seq>.npy (np.save of a list of frame dicts);
                read back by src/as_rigid_as_possible/main.py:58-74 `load_file`.
    Mesh-MNIST  src/mesh_mnist/add_laplacian.py:37-75 `process(sample)` -> sample dict; the list is written with
                np.save(open(path, 'wb'), list) (add_laplacian.py:85-86) and read by src/mesh_mnist/main.py:54-72.
    FAUST       no writer in the reference (the frames are distributed preprocessed); the reader is
                src/dense_correspondence/main.py:66-102 `read_data`, whose keys (V, F, L, D, DA as 0-d object arrays,
                label, label_inv, dist_mat) define the layout.

Both add_laplacian modules import `plyfile` (absent here) at module level, the ARAP one also lists its data directory on
import, so the modules cannot be imported.  The FUNCTIONS themselves only use utils.mesh / utils.graph / numpy / scipy:
they are taken from the reference source with `ast` at run time and executed unmodified (nothing is copied into this
repository) against the imported reference utils.  Input meshes are synthetic (the datasets are not downloadable).

Run ONLY in the build container:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_dataset_fixtures.py
"""
import ast
import os
import shutil
import sys
import tempfile
import types
import warnings

import numpy as np
import scipy as sp
import scipy.sparse  # noqa: F401
import torch

warnings.filterwarnings("ignore")
REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REF, "src"))
sys.path.insert(0, ROOT)
import utils.graph as RG  # noqa: E402  (reference)
import utils.mesh as RM  # noqa: E402
import utils.utils_pt as RU  # noqa: E402

from surfacenetworks_amd import datasets, mesh_ops  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def ref_function(rel_path, name, namespace):
    """The function `name` of a reference source file, compiled from its own text and bound to `namespace`."""
    path = os.path.join(REF, "src", rel_path)
    tree = ast.parse(open(path).read(), filename=path)
    node = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)
    code = compile(ast.Module(body=[node], type_ignores=[]), path, "exec")
    exec(code, namespace)
    return namespace[name]


def write_obj(path, V, F):
    with open(path, "w") as fh:
        for v in V:
            fh.write("v {:.9g} {:.9g} {:.9g}\n".format(*v))
        for f in F:
            fh.write("f {} {} {}\n".format(*(f + 1)))


rng = np.random.default_rng(2024)
work = tempfile.mkdtemp()
cwd = os.getcwd()
os.chdir(work)
try:
    # ---------------- ARAP ----------------
    os.makedirs("as_rigid_as_possible/data_obj")
    os.makedirs("as_rigid_as_possible/data_plus")
    for name, (n, m) in (("seq0", (6, 6)), ("seq1", (5, 5))):
        V0, F = mesh_ops.grid_cloth(n, m, rng)
        os.makedirs(f"as_rigid_as_possible/data_obj/{name}")
        for t in range(50):
            Vt = V0.copy()
            Vt[:, 2] += 0.03 * np.sin(7 * V0[:, 0] + 0.25 * t)
            Vt[:, 1] += 0.015 * np.cos(5 * V0[:, 1] + 0.2 * t)
            write_obj(f"as_rigid_as_possible/data_obj/{name}/{t:02d}.obj", Vt, F)
    ns = {"np": np, "sp": sp, "mesh": RM, "graph": RG, "utils": RU, "mypath": "as_rigid_as_possible/data_obj/", "print": lambda *a: None}
    process_arap = ref_function("as_rigid_as_possible/add_laplacian.py", "process", ns)
    for i, name in enumerate(("seq0", "seq1")):
        process_arap((i, name))                                              # writes data_plus/<name>.npy itself
        shutil.copy(f"as_rigid_as_possible/data_plus/{name}.npy", os.path.join(OUT, f"data_arap_{name}.npy"))
    # the reference's own reader on the same files vs the build's reader
    ns2 = {"np": np, "torch": torch, "args": types.SimpleNamespace(model="dir", dense=False)}
    load_file = ref_function("as_rigid_as_possible/main.py", "load_file", ns2)
    for name in ("seq0", "seq1"):
        p = os.path.join(OUT, f"data_arap_{name}.npy")
        ref_seq = load_file(p)
        mine = datasets.load_arap_sequence(p)
        assert len(ref_seq) == len(mine) == 50
        for t, (a, b) in enumerate(zip(ref_seq, mine)):
            assert np.array_equal(a["V"].numpy(), np.asarray(b["V"])) and np.array_equal(a["F"].numpy(), np.asarray(b["F"]))
            assert ("Di" in b) == (t < 10)
            if t < 10:
                assert abs(a["Di"] - b["Di"]).max() == 0 and abs(a["DiA"] - b["DiA"]).max() == 0 and a["L"] is None
                # and the build's own operator construction reproduces the stored operators bit for bit
                Di, DiA = mesh_ops.dirac(RM.load_obj(f"as_rigid_as_possible/data_obj/{name}/{t:02d}.obj")[0], np.asarray(b["F"]).astype(np.int64))
                assert abs(Di.astype(np.float32) - b["Di"]).max() == 0 and abs(DiA.astype(np.float32) - b["DiA"]).max() == 0
        print(f"ARAP {name}: {os.path.getsize(p) / 1024:.0f} KiB, reference load_file == datasets.load_arap_sequence")

    # ---------------- Mesh-MNIST ----------------
    ns3 = {"np": np, "sp": sp, "mesh": RM, "graph": RG, "print": lambda *a: None}
    process_mnist = ref_function("mesh_mnist/add_laplacian.py", "process", ns3)
    raw = []
    for k, nvert in enumerate((34, 41, 37, 33)):
        V, F = mesh_ops.delaunay_disc(nvert, rng)                            # unit-square-ish coordinates ...
        Vpix = (V - V.min(0)) / (V.max(0) - V.min(0) + 1e-9) * np.array([26.0, 26.0, 0.0]) + np.array([0.5, 0.5, 0.0])
        Vpix[:, 2] = 3.0 * rng.random(nvert)                                 # ... as 28x28-pixel coordinates with a height channel
        raw.append({"V": Vpix, "F": F.astype(np.int64), "label": int((3 * k + 1) % 10)})
    plus = [process_mnist((i, dict(s, V=s["V"].copy()))) for i, s in enumerate(raw)]
    p = os.path.join(OUT, "data_mnist_plus.np")
    np.save(open(p, "wb"), plus)                                             # as add_laplacian.py:85-86
    mine = datasets.load_mesh_mnist(p)
    assert len(mine) == 4 and all(set(s) == {"V", "F", "L", "flat_L", "Di", "DiA", "flat_Di", "flat_DiA", "label"} for s in mine)
    for a, b in zip(plus, mine):
        assert np.array_equal(a["V"], b["V"]) and abs(a["L"] - b["L"]).max() == 0 and abs(a["Di"] - b["Di"]).max() == 0 and a["label"] == b["label"]
        ops = mesh_ops.mesh_operators(np.asarray(b["V"], np.float64), np.asarray(b["F"]).astype(np.int64))
        # operators rebuilt from the STORED fp32 coordinates: same pattern, values to fp32 round-off of the coordinates
        assert np.array_equal(ops["L"].indices, b["L"].tocsr().indices) and abs(ops["Di"] - b["Di"]).max() <= 1e-4 * abs(b["Di"]).max()
    print(f"Mesh-MNIST: {os.path.getsize(p) / 1024:.0f} KiB, 4 samples")

    # ---------------- FAUST ----------------
    V, F = mesh_ops.torus_grid(7, 8, rng)
    d = RM.dist(V, F)
    a = RM.area(F, d)
    W, A = RM.cotangent_weights(F, a, d)
    L = (A * RG.laplacian(W, symmetric=False, normalized=False)).astype("float32")
    D, DA = RM.dirac(V, F)
    label = rng.permutation(V.shape[0])
    dist_mat = np.abs(rng.standard_normal((V.shape[0], V.shape[0]))).astype(np.float32)
    wrap = lambda m_: np.array(m_, dtype=object)
    p = os.path.join(OUT, "data_faust_frame.npz")
    np.savez(p, V=V, F=F, L=wrap(L), D=wrap(D.astype("float32")), DA=wrap(DA.astype("float32")), label=label,
             label_inv=np.argsort(label), dist_mat=dist_mat)
    ns4 = {"np": np, "torch": torch, "sp": sp, "utils": RU, "mesh": RM}
    read_data = ref_function("dense_correspondence/main.py", "read_data", ns4)
    _orig_load = np.load
    np.load = lambda f, *a_, **k_: _orig_load(f, *a_, **dict(k_, allow_pickle=True))       # (numpy >= 1.16.3 refuses pickles by default)
    try:
        ref_frame = read_data(p, types.SimpleNamespace(model="dir"))
    finally:
        np.load = _orig_load
    mine = datasets.load_faust_frame(p, device="cpu")
    assert torch.equal(ref_frame["V"], mine["V"]) and torch.equal(ref_frame["F"], mine["F"]) and torch.equal(ref_frame["G"], mine["G"])
    assert torch.equal(ref_frame["label"], mine["label"]) and torch.equal(ref_frame["label_inv"], mine["label_inv"])
    for k in ("Di", "DiA"):
        r = ref_frame[k]
        got = sp.sparse.coo_matrix((r._values().numpy(), (r._indices()[0].numpy(), r._indices()[1].numpy())), shape=tuple(r.shape)).tocsr()
        assert abs(got - mine[k]).max() == 0
    print(f"FAUST frame: {os.path.getsize(p) / 1024:.0f} KiB, reference read_data == datasets.load_faust_frame")
finally:
    os.chdir(cwd)
    shutil.rmtree(work, ignore_errors=True)
print("dataset fixtures written")
