"""One table of C-ABI calls — valid ones with their inputs, invalid ones with the status they must return — executed against
the HOST-POINTER TWINS (oracle/sn_host_twin.c, CPU suite) and against the DEVICE library (surfacenetworks_amd/libsn_hip.so,
-m gpu suite) with the same arguments: same statuses, bit-identical outputs."""
import ctypes as C

import numpy as np
import scipy.sparse as sp

SN_OK, SN_E_NULL, SN_E_SHAPE, SN_E_RANGE, SN_E_LD, SN_E_ALIGN, SN_E_WORKSPACE, SN_E_UNSUPPORTED = 0, -1, -2, -3, -4, -5, -6, -7
TWINS = ["sn_spmm_csr_f32", "sn_spmm_bsr4_f32", "sn_spmm_q3_f32", "sn_coo_to_csr_i32", "sn_csr_transpose_f32",
         "sn_blockdiag_concat_i32", "sn_blockdiag_concat_ragged_i32", "sn_validate_csr_i32", "sn_rb4_count", "sn_rb4_fill",
         "sn_spmm_rb4_f32", "sn_elu_into_f32", "sn_spmm_csr_ring_f32", "sn_csr_band_i32"]


class Backend:
    """host: numpy arrays passed by address to sn_host_*;  device: copies on cuda:0 passed to sn_*."""

    def __init__(self, kind):
        self.kind = kind
        if kind == "host":
            from oracle import c_oracle
            from surfacenetworks_amd import _lib

            self.lib = c_oracle.lib()
            for name in TWINS:
                fn = getattr(self.lib, name.replace("sn_", "sn_host_", 1))
                fn.restype, fn.argtypes = _lib.SIGNATURES[name]
            self.lib.sn_host_csr_transpose_workspace_bytes.restype = C.c_size_t
            self.lib.sn_host_csr_transpose_workspace_bytes.argtypes = [C.c_int64] * 3
        else:
            from surfacenetworks_amd import _lib

            self.lib = _lib.load()
        self.keep = []

    def fn(self, name):
        return getattr(self.lib, name if self.kind == "device" else name.replace("sn_", "sn_host_", 1))

    def buf(self, arr):
        """Pointer to a buffer holding `arr` (None -> NULL); fetch(handle) reads it back as numpy."""
        if arr is None:
            return None, None
        arr = np.ascontiguousarray(arr)
        if self.kind == "host":
            # 16-byte aligned copy (numpy only guarantees that for fresh allocations of sufficient size; make it explicit)
            raw = np.empty(arr.nbytes + 16, np.uint8)
            off = (-raw.ctypes.data) % 16
            view = raw[off: off + arr.nbytes].view(arr.dtype).reshape(arr.shape)
            view[...] = arr
            self.keep.append(raw)
            return view.ctypes.data, view
        import torch

        t = torch.from_numpy(arr.copy()).to("cuda")
        self.keep.append(t)
        return (t.data_ptr() if t.numel() else None), t

    def fetch(self, handle):
        if self.kind == "host":
            return np.array(handle)
        import torch

        torch.cuda.synchronize()
        return handle.cpu().numpy()


def _mesh_like_csr(M, K, rng, per_row=6):
    lens = rng.integers(0, per_row + 3, size=M)
    lens[0] = 0
    lens = np.minimum(lens, K)
    rows = np.repeat(np.arange(M), lens)
    cols = np.concatenate([np.sort(rng.choice(K, size=n, replace=False)) for n in lens]) if lens.sum() else np.zeros(0, np.int64)
    A = sp.csr_matrix((rng.standard_normal(len(rows)).astype(np.float32), (rows, cols)), shape=(M, K))
    A.sort_indices()
    return A


def run_valid(be: Backend):
    """Every twin entry point on well-formed inputs; returns {name: output arrays} for cross-backend comparison."""
    rng = np.random.default_rng(77)
    out = {}
    A = _mesh_like_csr(203, 150, rng)
    M, K, nnz, N = 203, 150, A.nnz, 32
    rp, _ = be.buf(A.indptr.astype(np.int32))
    ci, _ = be.buf(A.indices.astype(np.int32))
    va, _ = be.buf(A.data)
    x = rng.standard_normal((K, N)).astype(np.float32)
    xp, _ = be.buf(x)
    yp, yh = be.buf(np.full((M, N), np.nan, np.float32))
    assert be.fn("sn_spmm_csr_f32")(rp, ci, va, M, K, nnz, xp, N, 1, N, yp, N, 1, None) == SN_OK
    out["spmm_csr"] = be.fetch(yh)
    # validator, transpose
    fp, fh = be.buf(np.full(1, -1, np.int32))
    assert be.fn("sn_validate_csr_i32")(rp, ci, va, M, K, nnz, fp, None) == SN_OK
    out["validate"] = be.fetch(fh)
    trp, trh = be.buf(np.zeros(K + 1, np.int32))
    tcp, tch = be.buf(np.zeros(nnz, np.int32))
    tvp, tvh = be.buf(np.zeros(nnz, np.float32))
    from surfacenetworks_amd import _lib

    ws_bytes = 1 << 16
    wsp, _ = be.buf(np.zeros(ws_bytes, np.uint8))
    assert be.fn("sn_csr_transpose_f32")(rp, ci, va, M, K, nnz, trp, tcp, tvp, wsp, ws_bytes, None) == SN_OK
    out["transpose"] = [be.fetch(trh), be.fetch(tch), be.fetch(tvh)]
    # RB4: count, fill, product at N = 128
    Mb4 = (M + 3) // 4
    bpp, bph = be.buf(np.zeros(Mb4 + 1, np.int32))
    assert be.fn("sn_rb4_count")(rp, ci, M, K, bpp, wsp, ws_bytes, None) == SN_OK
    bcp, bch = be.buf(np.zeros(nnz, np.int32))
    bvp, bvh = be.buf(np.zeros((nnz, 4), np.float32))
    assert be.fn("sn_rb4_fill")(rp, ci, va, M, K, bpp, bcp, bvp, None) == SN_OK
    tot = int(be.fetch(bph)[-1])
    out["rb4"] = [be.fetch(bph), be.fetch(bch)[:tot], be.fetch(bvh)[:tot]]
    x128 = rng.standard_normal((K, 128)).astype(np.float32)
    x128p, _ = be.buf(x128)
    y128p, y128h = be.buf(np.full((M, 128), np.nan, np.float32))
    assert be.fn("sn_spmm_rb4_f32")(bpp, bcp, bvp, M, K, nnz, x128p, 128, 128, y128p, 128, None) == SN_OK
    out["spmm_rb4"] = be.fetch(y128h)
    # the sliding-window product on a square operator (some rows reach past the half window) and the band probe
    Ms = 700
    As = _mesh_like_csr(Ms, Ms, rng, per_row=7)
    As.sort_indices()
    srp, _ = be.buf(As.indptr.astype(np.int32))
    sci, _ = be.buf(As.indices.astype(np.int32))
    sva, _ = be.buf(As.data)
    xs = rng.standard_normal((Ms, 128)).astype(np.float32)
    xsp, _ = be.buf(xs)
    ysp, ysh = be.buf(np.full((Ms, 128), np.nan, np.float32))
    assert be.fn("sn_spmm_csr_ring_f32")(srp, sci, sva, Ms, Ms, As.nnz, xsp, 128, 128, ysp, 128, None) == SN_OK
    out["spmm_ring"] = be.fetch(ysh)
    bdp, bdh = be.buf(np.full(3, -1, np.int32))
    assert be.fn("sn_csr_band_i32")(srp, sci, Ms, Ms, bdp, None) == SN_OK
    out["band"] = be.fetch(bdh)
    # the same operator with the entries of every row reversed (columns descending): the probe must flag it — the ring kernel
    # bounds a row by its first and last entry — by reporting INT32_MAX as the longest row; the band is over all entries
    rev = np.concatenate([As.indices[As.indptr[r]:As.indptr[r + 1]][::-1] for r in range(Ms)]).astype(np.int32)
    rci, _ = be.buf(rev)
    bdp2, bdh2 = be.buf(np.full(3, -1, np.int32))
    assert be.fn("sn_csr_band_i32")(srp, rci, Ms, Ms, bdp2, None) == SN_OK
    out["band_unsorted"] = be.fetch(bdh2)
    assert out["band_unsorted"][1] == 0x7fffffff and out["band_unsorted"][0] == out["band"][0] and out["band"][1] < 64
    # block forms of a Dirac-like operator (4x4 blocks of M(p))
    Mb, Kb, per = 40, 30, 3
    brp_ = np.arange(0, (Mb + 1) * per, per, dtype=np.int32)
    bcol = np.concatenate([np.sort(rng.choice(Kb, per, replace=False)) for _ in range(Mb)]).astype(np.int32)
    p = rng.standard_normal((Mb * per, 3)).astype(np.float32)
    z = np.zeros(Mb * per, np.float32)
    blocks = np.stack([np.stack([z, p[:, 0], p[:, 1], p[:, 2]], 1), np.stack([-p[:, 0], z, p[:, 2], -p[:, 1]], 1),
                       np.stack([-p[:, 1], -p[:, 2], z, p[:, 0]], 1), np.stack([-p[:, 2], p[:, 1], -p[:, 0], z], 1)], 1).astype(np.float32)
    q = np.zeros((Mb * per, 4), np.float32)
    q[:, :3] = p
    q[:, 3] = bcol.view(np.float32)
    xq = rng.standard_normal((Kb, 4 * N)).astype(np.float32)
    xqp, _ = be.buf(xq)
    b_rp, _ = be.buf(brp_)
    b_ci, _ = be.buf(bcol)
    b_va, _ = be.buf(blocks.reshape(-1))
    qp, _ = be.buf(q)
    for name, call in (("spmm_bsr4", lambda y: be.fn("sn_spmm_bsr4_f32")(b_rp, b_ci, b_va, Mb, Kb, Mb * per, xqp, 4 * N, 4, N, y, 4 * N, 4, None)),
                       ("spmm_q3", lambda y: be.fn("sn_spmm_q3_f32")(b_rp, qp, Mb, Kb, Mb * per, xqp, 4 * N, 4, N, y, 4 * N, 4, None))):
        yq, yqh = be.buf(np.full((Mb, 4 * N), np.nan, np.float32))
        assert call(yq) == SN_OK
        out[name] = be.fetch(yqh)
    # COO -> CSR with interior empty rows, batched
    B, R, Kc = 3, 5, 4
    ent = np.array([[0, 0, 1], [0, 0, 3], [0, 3, 0], [1, 1, 2], [2, 0, 0], [2, 4, 3]], np.int64)
    ibp, _ = be.buf(ent[:, 0].copy())
    irp, _ = be.buf(ent[:, 1].copy())
    icp, _ = be.buf(ent[:, 2].copy())
    orp, orh = be.buf(np.zeros(B * R + 1, np.int32))
    ocp, och = be.buf(np.zeros(len(ent), np.int32))
    assert be.fn("sn_coo_to_csr_i32")(ibp, irp, icp, len(ent), B, R, Kc, orp, ocp, None) == SN_OK
    out["coo_to_csr"] = [be.fetch(orh), be.fetch(och)]
    # pooled batch assembly, padded and ragged, CSR entries
    mats = [_mesh_like_csr(7, 5, rng, 3), _mesh_like_csr(4, 6, rng, 3), _mesh_like_csr(9, 3, rng, 2)]
    prp = np.concatenate([m.indptr for m in mats]).astype(np.int32)
    pci = np.concatenate([m.indices for m in mats]).astype(np.int32)
    pva = np.concatenate([m.data for m in mats]).astype(np.float32)
    rp_off = np.cumsum([0] + [m.shape[0] + 1 for m in mats])
    e_off = np.cumsum([0] + [m.nnz for m in mats])
    sel = [2, 0, 1, 0]
    cnt = [mats[i].nnz for i in sel]
    o_off = np.cumsum([0] + cnt)
    total = int(o_off[-1])
    prpp, _ = be.buf(prp)
    pcip, _ = be.buf(pci)
    pvap, _ = be.buf(pva)
    s0, s1 = 10, 7
    d4, _ = be.buf(np.array([[rp_off[i], e_off[i], mats[i].shape[0], o_off[j]] for j, i in enumerate(sel)], np.int64))
    a1, a1h = be.buf(np.zeros(len(sel) * s0 + 1, np.int32))
    a2, a2h = be.buf(np.zeros(total, np.int32))
    a3, a3h = be.buf(np.zeros(total, np.float32))
    assert be.fn("sn_blockdiag_concat_i32")(prpp, pcip, pvap, d4, len(sel), s0, s1, total, 1, a1, a2, a3, None) == SN_OK
    out["concat"] = [be.fetch(a1h), be.fetch(a2h), be.fetch(a3h)]
    ro = np.cumsum([0] + [mats[i].shape[0] for i in sel])
    co = np.cumsum([0] + [mats[i].shape[1] for i in sel])
    d6, _ = be.buf(np.array([[rp_off[i], e_off[i], mats[i].shape[0], o_off[j], ro[j], co[j]] for j, i in enumerate(sel)], np.int64))
    r1, r1h = be.buf(np.zeros(int(ro[-1]) + 1, np.int32))
    r2, r2h = be.buf(np.zeros(total, np.int32))
    r3, r3h = be.buf(np.zeros(total, np.float32))
    assert be.fn("sn_blockdiag_concat_ragged_i32")(prpp, pcip, pvap, d6, len(sel), int(ro[-1]), int(co[-1]), total, 1, r1, r2, r3, None) == SN_OK
    out["concat_ragged"] = [be.fetch(r1h), be.fetch(r2h), be.fetch(r3h)]
    want = sp.block_diag([mats[i] for i in sel], format="csr")
    want.sort_indices()
    assert np.array_equal(out["concat_ragged"][0], want.indptr) and np.array_equal(out["concat_ragged"][1], want.indices)
    # ELU into a strided destination
    src = (rng.standard_normal((11, 8)) * 2).astype(np.float32)
    sp_, _ = be.buf(src)
    dp, dh = be.buf(np.full((11, 16), np.nan, np.float32))
    assert be.fn("sn_elu_into_f32")(sp_, 8, dp, 16, 11, 8, None) == SN_OK
    out["elu"] = be.fetch(dh)
    out["elu_src"] = src
    return out


def invalid_calls(be: Backend):
    """(description, status returned, status expected) for calls that must be refused before anything is touched."""
    big = 2**31
    one, _ = be.buf(np.zeros(64, np.float32))          # a valid, 16-byte aligned pointer for the slots that need one
    onei, _ = be.buf(np.zeros(64, np.int32))
    f = be.fn
    calls = [
        ("csr: negative M", f("sn_spmm_csr_f32")(onei, onei, one, -1, 4, 4, one, 32, 1, 32, one, 32, 1, None), SN_E_SHAPE),
        ("csr: N < 1", f("sn_spmm_csr_f32")(onei, onei, one, 1, 4, 4, one, 32, 1, 0, one, 32, 1, None), SN_E_SHAPE),
        ("csr: M past int32", f("sn_spmm_csr_f32")(onei, onei, one, big, 4, 4, one, 32, 1, 32, one, 32, 1, None), SN_E_RANGE),
        ("csr: nnz past int32", f("sn_spmm_csr_f32")(onei, onei, one, 1, 4, big, one, 32, 1, 32, one, 32, 1, None), SN_E_RANGE),
        ("csr: M == 0 is a no-op", f("sn_spmm_csr_f32")(None, None, None, 0, 4, 0, None, 32, 1, 32, None, 32, 1, None), SN_OK),
        ("csr: null rowptr", f("sn_spmm_csr_f32")(None, onei, one, 1, 4, 1, one, 32, 1, 32, one, 32, 1, None), SN_E_NULL),
        ("csr: null Y", f("sn_spmm_csr_f32")(onei, onei, one, 1, 4, 1, one, 32, 1, 32, None, 32, 1, None), SN_E_NULL),
        ("csr: ld < N", f("sn_spmm_csr_f32")(onei, onei, one, 1, 4, 1, one, 32, 1, 32, one, 16, 1, None), SN_E_LD),
        ("csr: group 3", f("sn_spmm_csr_f32")(onei, onei, one, 1, 4, 1, one, 32, 1, 32, one, 32, 3, None), SN_E_LD),
        ("csr: group-4 ld < 4N", f("sn_spmm_csr_f32")(onei, onei, one, 4, 4, 1, one, 64, 4, 32, one, 128, 4, None), SN_E_LD),
        ("bsr4: N = 24", f("sn_spmm_bsr4_f32")(onei, onei, one, 1, 1, 1, one, 96, 4, 24, one, 96, 4, None), SN_E_UNSUPPORTED),
        ("bsr4: 4*Mb+1 past int32", f("sn_spmm_bsr4_f32")(onei, onei, one, big // 4, 1, 1, one, 128, 4, 32, one, 128, 4, None), SN_E_RANGE),
        ("bsr4: ld % 4", f("sn_spmm_bsr4_f32")(onei, onei, one, 1, 1, 1, one, 130, 4, 32, one, 128, 4, None), SN_E_ALIGN),
        ("q3: null records", f("sn_spmm_q3_f32")(onei, None, 1, 1, 1, one, 128, 4, 32, one, 128, 4, None), SN_E_NULL),
        ("q3: N = 8", f("sn_spmm_q3_f32")(onei, one, 1, 1, 1, one, 32, 4, 8, one, 32, 4, None), SN_E_UNSUPPORTED),
        ("rb4: N = 32", f("sn_spmm_rb4_f32")(onei, onei, one, 4, 4, 4, one, 32, 32, one, 32, None), SN_E_UNSUPPORTED),
        ("rb4: M past int32", f("sn_spmm_rb4_f32")(onei, onei, one, big, 4, 4, one, 128, 128, one, 128, None), SN_E_RANGE),
        ("ring: not square", f("sn_spmm_csr_ring_f32")(onei, onei, one, 4, 5, 4, one, 128, 128, one, 128, None), SN_E_UNSUPPORTED),
        ("ring: N = 32", f("sn_spmm_csr_ring_f32")(onei, onei, one, 4, 4, 4, one, 32, 32, one, 32, None), SN_E_UNSUPPORTED),
        ("ring: M past int32", f("sn_spmm_csr_ring_f32")(onei, onei, one, big, big, 4, one, 128, 128, one, 128, None), SN_E_RANGE),
        ("ring: null rowptr", f("sn_spmm_csr_ring_f32")(None, onei, one, 4, 4, 4, one, 128, 128, one, 128, None), SN_E_NULL),
        ("ring: ld % 4", f("sn_spmm_csr_ring_f32")(onei, onei, one, 4, 4, 4, one, 130, 128, one, 128, None), SN_E_ALIGN),
        ("ring: M == 0 is a no-op", f("sn_spmm_csr_ring_f32")(None, None, None, 0, 0, 0, None, 128, 128, None, 128, None), SN_OK),
        ("band: null output", f("sn_csr_band_i32")(onei, onei, 4, 4, None, None), SN_E_NULL),
        ("band: M past int32", f("sn_csr_band_i32")(onei, onei, big, 4, onei, None), SN_E_RANGE),
        ("coo: B < 1", f("sn_coo_to_csr_i32")(None, None, None, 0, 0, 4, 4, onei, onei, None), SN_E_SHAPE),
        ("coo: B*R past int32", f("sn_coo_to_csr_i32")(None, None, None, 0, 4096, 2**20, 4, onei, onei, None), SN_E_RANGE),
        ("coo: null rowptr", f("sn_coo_to_csr_i32")(None, None, None, 0, 1, 4, 4, None, onei, None), SN_E_NULL),
        ("transpose: workspace too small", f("sn_csr_transpose_f32")(onei, onei, one, 4, 1000, 4, onei, onei, one, one, 8, None), SN_E_WORKSPACE),
        ("concat: vals_per_entry 3", f("sn_blockdiag_concat_i32")(onei, onei, one, one, 1, 4, 4, 4, 3, onei, onei, one, None), SN_E_UNSUPPORTED),
        ("concat: B*size0 past int32", f("sn_blockdiag_concat_i32")(onei, onei, one, one, 1024, 4 * 600_000, 4, 4, 1, onei, onei, one, None), SN_E_RANGE),
        ("ragged: rows past int32", f("sn_blockdiag_concat_ragged_i32")(onei, onei, one, one, 1, big, 4, 4, 1, onei, onei, one, None), SN_E_RANGE),
        ("ragged: B == 0 with rows", f("sn_blockdiag_concat_ragged_i32")(onei, onei, one, one, 0, 5, 4, 0, 1, onei, onei, one, None), SN_E_SHAPE),
        ("validate: null flags", f("sn_validate_csr_i32")(onei, onei, one, 1, 4, 1, None, None), SN_E_NULL),
        ("elu: ld < C", f("sn_elu_into_f32")(one, 4, one, 8, 2, 8, None), SN_E_SHAPE),
        ("elu: null src", f("sn_elu_into_f32")(None, 8, one, 8, 2, 8, None), SN_E_NULL),
    ]
    return calls
