"""LazySparse (resident.py) keeps every promise of the tensors the reference's batching functions return
(src/utils/utils_pt.py:21-69) — host logic, no GPU: without one the batching functions take the reference's host path and
the lazy handles materialise to exactly the reference's tensors."""
import numpy as np
import scipy.sparse as sp
import torch

import surfacenetworks_amd.utils_pt as U
from surfacenetworks_amd.resident import LazySparse, reference_cat, reference_diag_cat, reference_sp_to_coo


def _mats():
    rng = np.random.default_rng(0)
    return [sp.random(12, 9, density=0.3, format=f, dtype=np.float32, random_state=int(rng.integers(1 << 30)))
            for f in ("csr", "coo", "csc")]


def test_handle_of_a_scipy_matrix_behaves_like_the_references_tensor():
    for A in _mats():
        want = reference_sp_to_coo(A)
        h = U.sp_sparse_to_pt_sparse(A)
        assert isinstance(h, LazySparse) and isinstance(h, torch.Tensor)
        assert h.layout == torch.sparse_coo and h.is_sparse and not h.is_cuda and h.device.type == "cpu"
        assert h.size() == want.size() and h.size(1) == 9 and h.dim() == 2 and h.shape == want.shape and h.dtype == want.dtype
        assert not h.is_coalesced() and not h.requires_grad
        assert not h._sn_payload.real                                   # nothing was built for the metadata
        assert torch.equal(h._indices(), want._indices()) and torch.equal(h._values(), want._values())
        assert h._nnz() == want._nnz()
        c = h.coalesce()
        assert c.is_coalesced() and torch.equal(c._indices(), want.coalesce()._indices())
        assert torch.equal(h.to_dense(), want.to_dense())
        x = torch.randn(9, 5)
        assert torch.equal(torch.mm(h, x), torch.mm(want, x)) and torch.equal(torch.sparse.mm(c, x), torch.sparse.mm(want.coalesce(), x))
        assert torch.equal((h + h).to_dense(), (want + want).to_dense())
        assert torch.equal(h.t().to_dense(), want.t().to_dense())
        assert h.detach().shape == h.shape and h.cpu() is h and h.to("cpu") is h
        assert h.double().dtype == torch.float64
    A64 = sp.random(5, 5, density=0.5, format="csr", dtype=np.float64, random_state=1)
    assert U.sp_sparse_to_pt_sparse(A64).dtype == torch.float64                      # dtype kept, as utils_pt.py:56-69


def test_batching_functions_without_a_gpu_are_the_references():
    mats = _mats()
    hs = [U.sp_sparse_to_pt_sparse(A) for A in mats]
    real = [reference_sp_to_coo(A) for A in mats]
    for fn, ref in ((U.sparse_diag_cat, reference_diag_cat), (U.sparse_cat, reference_cat)):
        got, want = fn(hs, 14, 10), ref(real, 14, 10)
        assert got.is_coalesced() and got.shape == want.shape
        assert torch.equal(got._indices(), want._indices()) and torch.equal(got._values(), want._values())
        mixed = fn([hs[0], real[1], hs[2]], 14, 10)
        assert torch.equal(mixed.to_dense(), want.to_dense())


def test_cache_signature_notices_replaced_arrays_and_in_place_edits():
    import scipy.sparse as sp

    from surfacenetworks_amd.resident import ResidentCache

    A = sp.random(40, 40, 0.2, "csr", np.float32, random_state=1)
    s0 = ResidentCache._signature(A)
    assert ResidentCache._signature(A) == s0                       # asked twice: the same
    A.data *= 2                                                    # the same arrays, new values
    s1 = ResidentCache._signature(A)
    assert s1 != s0
    A.data = A.data.copy()                                         # the same values, a new array
    assert ResidentCache._signature(A) != s1
    E = sp.csr_matrix((4, 4), dtype=np.float32)                    # no entries: nothing to probe
    assert ResidentCache._signature(E) == ResidentCache._signature(E)


def test_an_edit_that_misses_every_probe_changes_the_full_checksum():
    """ResidentCache._signature probes eight values per array (O(1) per call); the round-robin full checksum (_digest) is what
    makes the staleness check deterministic: an edit between the probes leaves the signature and changes the digest."""
    import scipy.sparse as sp

    from surfacenetworks_amd.resident import ResidentCache, _digest

    A = sp.random(200, 200, 0.2, "csr", np.float32, random_state=2)
    s0, d0 = ResidentCache._signature(A), _digest(A)
    step = max(1, A.data.size // 8)
    A.data[1] += 1.0                                               # index 1 is not one of 0, step, 2 step, ...
    assert step > 1 and ResidentCache._signature(A) == s0 and _digest(A) != d0
    d1 = _digest(A)
    j = 3 if (3 % max(1, A.indices.size // 8)) else 4
    A.indices[j], A.indices[j + 1] = A.indices[j + 1], A.indices[j]           # a structural edit that keeps nnz
    assert _digest(A) != d1
