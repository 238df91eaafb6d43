"""Host-pointer twins of the C-ABI (oracle/sn_host_twin.c, test infrastructure): same signatures, same argument checks and
status codes as the device entry points, oracle arithmetic.  CPU suite: the twins against the oracle and the status table;
the -m gpu counterpart (tests/test_spmm_gpu.py::test_device_library_matches_host_twins) runs the SAME table on the device."""
import numpy as np

import abi_cases as ac


def test_every_twin_mirrors_a_declared_entry_point():
    import os
    import re

    from surfacenetworks_amd import _lib

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "sn_spmm.h")).read()
    twin_src = open(os.path.join(root, "oracle", "sn_host_twin.c")).read()
    twins = set(re.findall(r"^(?:int|size_t) (sn_host_\w+)\(", twin_src, flags=re.M))
    assert {t.replace("sn_host_", "sn_", 1) for t in twins if not t.endswith("_workspace_bytes")} == set(ac.TWINS)
    for name in ac.TWINS:
        assert re.search(rf"\b{name}\(", header) and name in _lib.SIGNATURES


def test_twins_reject_invalid_calls_with_the_documented_status():
    for what, got, want in ac.invalid_calls(ac.Backend("host")):
        assert got == want, (what, got, want)


def test_twins_compute_the_oracle_results():
    from oracle import c_oracle

    out = ac.run_valid(ac.Backend("host"))
    assert out["validate"][0] == 0 and not np.isnan(out["spmm_csr"]).any()
    assert np.array_equal(out["spmm_bsr4"], out["spmm_q3"])          # packed quaternion records == explicit 4x4 blocks
    # RB4 product == CSR product of the same operator (other N, so compare through the oracle)
    rng = np.random.default_rng(77)
    A = ac._mesh_like_csr(203, 150, rng)
    x = rng.standard_normal((150, 32)).astype(np.float32)
    assert np.array_equal(out["spmm_csr"], c_oracle.spmm_csr(A.indptr, A.indices, A.data, x.ravel(), 32).reshape(203, 32))
    tr = c_oracle.csr_transpose(A.indptr, A.indices, A.data, 150)
    for a, b in zip(out["transpose"], tr):
        assert np.array_equal(a, b)
    assert out["coo_to_csr"][0].tolist() == [0, 2, 2, 2, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 6]
    assert np.isnan(out["elu"][:, 8:]).all()                          # the strided destination's other half is untouched
    assert np.allclose(out["elu"][:, :8], c_oracle.elu(out["elu_src"]), rtol=1e-6, atol=1e-7)


def test_ring_twin_is_the_csr_product_and_the_band_of_the_operator():
    out = ac.run_valid(ac.Backend("host"))
    assert not np.isnan(out["spmm_ring"]).any()
    band, longest, outside = (int(v) for v in out["band"])
    assert band > 160 and longest >= 7 and 0 < outside <= 700          # a 700-row operator with columns anywhere: rows past the window
