"""The drop-in boundary: the C-ABI library loads and exports every symbol include/sn_spmm.h declares, the ctypes
binding lists exactly those, the product has no route to the oracle or to a CPU fallback, and the Python operator API
has the reference's names.  No GPU needed (no compute calls)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "surfacenetworks_amd")


def header_symbols():
    src = open(os.path.join(ROOT, "include", "sn_spmm.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sn_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from surfacenetworks_amd import _lib

    syms = header_symbols()
    assert len(syms) >= 12
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/sn_spmm.h but not exported by libsn_hip.so"
    assert sorted(_lib.SIGNATURES) == syms, "ctypes binding and header disagree"
    assert _lib.load().sn_abi_version() == 1
    assert b"int32" in _lib.load().sn_status_string(-3)


def test_missing_library_fails_loudly(monkeypatch):
    from surfacenetworks_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", os.path.join(PKG, "does_not_exist.so"))
    with pytest.raises(RuntimeError, match="no CPU/eager fallback"):
        _lib.load()


def test_product_never_touches_oracle_or_torch_sparse_mm():
    """Static scan: nothing under surfacenetworks_amd/ imports oracle/, and the hot path never calls torch's own
    sparse matmul (torch.mm / torch.sparse.mm / torch.spmm on a sparse operand) as a fallback."""
    bad = []
    for dirpath, _, files in os.walk(PKG):
        for fn in files:
            if not fn.endswith((".py", ".hip", ".cpp", ".h")):
                continue
            text = open(os.path.join(dirpath, fn)).read()
            code = re.sub(r'""".*?"""', "", text, flags=re.S)
            code = re.sub(r"#.*", "", code)
            if re.search(r"^\s*(from|import)\s+oracle\b", code, flags=re.M) or "oracle." in code or "libsn_oracle" in code:
                bad.append((fn, "oracle"))
            if re.search(r"torch\.(sparse\.mm|spmm|sparse\.addmm|smm)\b", code) or re.search(r"torch\.mm\(", code):
                bad.append((fn, "torch sparse mm"))
    assert not bad, bad


def test_cpu_tensors_are_rejected_not_computed():
    from surfacenetworks_amd import functional as snF
    from surfacenetworks_amd.operators import SparseOperator

    op = SparseOperator(torch.tensor([0, 1, 2], dtype=torch.int32), torch.tensor([0, 1], dtype=torch.int32),
                        torch.ones(2), (2, 2))
    with pytest.raises(RuntimeError, match="no CPU"):
        snF.spmm(op, torch.ones(2, 4))
    A = torch.sparse_coo_tensor(torch.tensor([[0, 1], [0, 1]]), torch.ones(2), (2, 2)).coalesce()
    with pytest.raises(RuntimeError, match="no CPU"):
        snF.spmm(A, torch.ones(2, 4))


def test_reference_operator_api_names_present():
    """Public interface of the reference's L3 layer: same names, so `import ... as utils` is the swap."""
    import surfacenetworks_amd.utils_pt as U

    for name in ["GraphConv1x1", "GraphBatchNorm", "LapResNet2", "DenseLapResNet2", "DirResNet2", "AvgResNet2",
                 "MlpResNet2", "global_average", "sparse_cat", "sparse_diag_cat", "sp_sparse_to_pt_sparse",
                 "to_dense_batched"]:
        assert hasattr(U, name), name


def test_state_dict_schema_matches_reference():
    """219 entries for the ARAP DirModel with the reference's key names and shapes."""
    from oracle import ref_blocks as OB
    from surfacenetworks_amd import arap, dense_correspondence, mesh_mnist

    pairs = [(arap.DirModel(), OB.ArapDirModel()), (arap.Model(15), OB.ArapLapModel(15)), (mesh_mnist.Model(), OB.MnistLapModel()),
             (mesh_mnist.DirModel(), OB.MnistDirModel()), (dense_correspondence.SiameseModel("lap", 15), OB.SiameseModel("lap", 15))]
    for mine, ref in pairs:
        a, b = mine.state_dict(), ref.state_dict()
        assert list(a) == list(b)
        assert all(a[k].shape == b[k].shape for k in a)
    sd = arap.DirModel().state_dict()
    assert len(sd) == 219
    assert sd["conv1.fc.weight"].shape == (128, 6) and sd["rn0.bn_fc0.fc.weight"].shape == (128, 256)
    assert sd["rn14.bn_fc1.bn.running_var"].shape == (256,) and sd["conv2.fc.weight"].shape == (120, 128)
    assert sum(p.numel() for p in arap.DirModel().parameters()) == 1018872
    assert sum(p.numel() for p in mesh_mnist.Model().parameters()) == 90314


def test_debug_validation_mode_rejects_malformed_operators(cpu_kernels, monkeypatch):
    """SN_DEBUG_VALIDATE=1: SparseOperator checks the CSR arrays it is built from (host logic; the device kernel itself is
    tested in tests/test_fullsize_gpu.py)."""
    import torch

    from surfacenetworks_amd import operators

    rp = torch.tensor([0, 2, 3], dtype=torch.int32)
    ci = torch.tensor([0, 7, 1], dtype=torch.int32)
    va = torch.ones(3)
    op = operators.SparseOperator(rp, ci, va, (2, 4))
    with pytest.raises(ValueError, match="column index out of range"):
        op.validate()
    operators.SparseOperator(rp, torch.tensor([0, 3, 1], dtype=torch.int32), va, (2, 4)).validate()


def test_environment_switches_are_the_documented_ones():
    """Every environment variable the library (getenv) and the package (os.environ) read is listed — completely — in
    include/sn_spmm.h ("SWITCHES:" line), and nothing listed is dead.  Superseded kernels and their A/B toggles are deleted,
    not hidden behind undocumented switches."""
    header = open(os.path.join(ROOT, "include", "sn_spmm.h")).read()
    documented = set(re.search(r"SWITCHES:((?:\s+SN_[A-Z0-9_]+)+)", header).group(1).split())
    found = set()
    for dirpath, _, files in os.walk(PKG):
        for fn in files:
            if not fn.endswith((".py", ".hip", ".h")):
                continue
            text = open(os.path.join(dirpath, fn)).read()
            found.update(re.findall(r'getenv\(\s*"(SN_[A-Z0-9_]+)"', text))
            found.update(re.findall(r'environ(?:\.get|\.setdefault)?\(\s*"(SN_[A-Z0-9_]+)"', text))
            found.update(re.findall(r'environ\[\s*"(SN_[A-Z0-9_]+)"', text))
            # compile-time ablation toggles of earlier rounds (SN_X_*) are gone too
            assert not re.search(r"\bSN_X_[A-Z0-9_]+", text), f"{fn} still carries an SN_X_* ablation toggle"
    assert found == documented, (sorted(found - documented), sorted(documented - found))
