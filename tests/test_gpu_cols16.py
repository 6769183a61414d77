"""16-bit column stream of the SELL-64 SpMV (ox_sell_compress_cols): the codes decode to the
int32 columns exactly, the SpMV is bit-identical with and without them, and slices that need a
third base fall back to their int32 columns."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _decode(P):
    """Host decode of (cols16, cbase); returns (cols, compressed mask per slot)."""
    code = P.cols16.cpu().numpy().view(np.uint16).astype(np.int64)
    cb = P.cbase.cpu().numpy().reshape(-1, 2).astype(np.int64)
    pair = np.arange(P.size) // 128
    sp = P.slice_ptr.cpu().numpy()
    first = cb[sp[:-1] // 128, 0]  # < 0: the slice keeps its int32 columns
    slot_slice = np.searchsorted(sp, np.arange(P.size), side="right") - 1
    comp = first[slot_slice] >= 0
    dec = np.where(code >= 32768, cb[pair, 1], cb[pair, 0]) + (code & 0x7FFF)
    return dec, comp


@pytest.mark.parametrize("dim,N,deg", [(2, 24, 2), (3, 10, 2), (3, 20, 1)])
def test_codes_decode_to_the_int32_columns(dim, N, deg):
    from oasisx_amd import fem
    from oasisx_amd.la import SellMatrix
    from tests.helpers import tg_mesh

    V = fem.FunctionSpace(tg_mesh(dim, N), deg)
    P = V.pattern
    SellMatrix(P)  # builds the stream
    assert P.cols16 is not None and 0.0 < P.frac16 <= 1.0
    dec, comp = _decode(P)
    cols = P.cols.cpu().numpy().astype(np.int64)
    assert comp.mean() == pytest.approx(P.frac16)
    np.testing.assert_array_equal(dec[comp], cols[comp])


def _random_pattern(n, row_cols, device):
    """SELL pattern from explicit per-row column lists."""
    from oasisx_amd import fem

    keys, lens = [], []
    for r, cs in enumerate(row_cols):
        cs = np.unique(np.asarray(cs, dtype=np.int64))
        keys.append(r * n + cs)
        lens.append(len(cs))
    tail = np.arange(len(row_cols), n, dtype=np.int64)  # remaining rows: diagonal only
    keys = torch.as_tensor(np.concatenate(keys + [tail * n + tail]), device=device)
    row_len = torch.as_tensor(np.concatenate([np.asarray(lens, dtype=np.int64), np.ones(len(tail), dtype=np.int64)]),
                              device=device)
    row_ptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    row_ptr[1:] = torch.cumsum(row_len, 0)
    return fem.build_sell(n, n, keys, row_len, row_ptr)


def test_three_groups_fall_back_and_spmv_is_bit_identical():
    from oasisx_amd import _lib
    from oasisx_amd.la import SellMatrix

    dev = torch.device("cuda")
    rng = np.random.default_rng(7)
    n = 64 * 6
    big = 400000
    nn = big + n
    rows = []
    for r in range(n):
        s = r // 64
        cs = [r, (r + 1) % n]
        if s == 1:  # two groups: near + one far block -> still 16 bit
            cs += [100000 + r]
        if s == 2:  # three groups in one storage pair for some lanes -> int32 fallback
            cs += [100000 + r] if r % 2 else [300000 + r]
            cs += [200000 + r]
        if s == 3:  # one group wider than 15 bits but narrower than two -> second base
            cs += [r + 40000]
        rows.append(cs)
    P = _random_pattern(nn, rows, dev)
    A = SellMatrix(P)
    slot_row, slot_k = P.slot_rows_k()
    rl = np.zeros(P.n_slices * 64, dtype=np.int64)
    rl[:nn] = P.row_len.cpu().numpy()
    vals = np.where(slot_k < rl[slot_row], rng.uniform(0.5, 1.5, P.size), 0.0)  # padding slots hold 0
    A.vals.copy_(torch.as_tensor(vals, device=dev))
    dec, comp = _decode(P)
    cols = P.cols.cpu().numpy().astype(np.int64)
    np.testing.assert_array_equal(dec[comp], cols[comp])
    sp = P.slice_ptr.cpu().numpy()
    first = P.cbase.cpu().numpy().reshape(-1, 2)[sp[:-1] // 128, 0]
    assert first[2] < 0, "the three-group slice must keep its int32 columns"
    assert (first[[0, 1, 3, 4, 5]] >= 0).all()
    lib = _lib.load()
    for nc in (1, 3):
        x = torch.as_tensor(rng.standard_normal((nn, nc)), device=dev).contiguous()
        ys = []
        for var in (1, 3):
            A.set_levels(var)
            y = torch.zeros_like(x)
            A.mult(x, y, nc)
            ys.append(y.cpu().numpy())
        A.set_levels(None)
        np.testing.assert_array_equal(ys[0], ys[1])
        ref = P.to_csr(A.vals) @ x.cpu().numpy()
        np.testing.assert_allclose(ys[1], ref, rtol=1e-13, atol=1e-13)


def test_value_dictionary_is_bit_identical_and_drops_when_values_change():
    """la.SellMatrix.freeze: <= 256 distinct values -> 1-byte codes; the SpMV and the Krylov solve are
    bit-identical with and without them; changing the values afterwards drops the codes."""
    import ctypes as C

    from oasisx_amd import _lib
    from oasisx_amd.ksp import KSPSolver
    from tests.helpers import make_hip_problem

    S, clock, mesh = make_hip_problem(3, 8, u_deg=2)
    lib = _lib.load()
    for A in (S._M, S._Ap):
        assert A.vcode is not None and 1 <= A._struct.n_dict <= 256, "box meshes have few distinct values"
        np.testing.assert_array_equal(A.vdict[A.vcode.long()].cpu().numpy().view(np.int64),
                                      A.vals.cpu().numpy().view(np.int64))
        n = A.pattern.n_cols
        for nc in (1, 3):
            x = torch.randn(n, nc, dtype=torch.float64, device="cuda").contiguous()
            ys = []
            for var in (3, 7, 15):
                A.set_levels(var)
                y = torch.zeros(A.pattern.n_rows, nc, dtype=torch.float64, device="cuda")
                A.mult(x, y, nc)
                ys.append(y.cpu().numpy())
            A.set_levels(None)
            np.testing.assert_array_equal(ys[0].view(np.int64), ys[1].view(np.int64))
            np.testing.assert_array_equal(ys[0].view(np.int64), ys[2].view(np.int64))
    # the assembled convective matrix has no small dictionary
    S.assemble_first(0.005, 0.01)
    assert not S._A.freeze()
    # values change after freeze(): the codes must not be used any more
    M = S._M
    rows = torch.tensor([0, 5], dtype=torch.int32, device="cuda")
    M.zero_rows(rows, 1.0)
    M.ref()
    assert M.vcode is None and M._struct.n_dict == 0 and not M._struct.vcode
    x = torch.randn(M.pattern.n_cols, 1, dtype=torch.float64, device="cuda")
    y = torch.zeros(M.pattern.n_rows, 1, dtype=torch.float64, device="cuda")
    M.mult(x, y, 1)
    np.testing.assert_allclose(y.cpu().numpy(), M.to_scipy() @ x.cpu().numpy(), rtol=1e-13, atol=1e-15)


def test_rectangular_operators_with_value_codes_are_bit_identical():
    """la.MultiSellMatrix.freeze: the pre-assembled P / G / D operators (reference
    fracstep.py:392-404) through packed 1-byte codes + 16-bit columns == through f64 + int32."""
    import ctypes as C

    from oasisx_amd import _lib
    from tests.helpers import make_hip_problem

    # N = 8: h = 1/4 is exact in binary, so all cells have bit-identical geometry and the operators
    # few distinct values (with h = 1/3 the rounding noise of the vertex coordinates alone makes
    # thousands of distinct bit patterns and freeze() declines -- checked below)
    S, clock, mesh = make_hip_problem(3, 8, u_deg=2, low_memory=False)
    lib = _lib.load()
    nu_, nq = S._n_u, S._n_q
    for Mat, v2s, nx, ny in ((S._p_vdxi_Mat, False, nq, nu_ * 3), (S._grad_p_Mat, False, nq, nu_ * 3),
                             (S._divu_Mat, True, nu_ * 3, nq)):
        assert Mat.vcode is not None and 1 <= Mat._struct.n_dict <= 256
        # the codes reproduce the values bit for bit
        code = Mat.vcode.cpu().numpy().astype(np.int64)
        dec = np.stack([Mat.vdict.cpu().numpy()[(code >> (8 * d)) & 255] for d in range(3)], axis=1)
        np.testing.assert_array_equal(dec.view(np.int64), Mat.vals.cpu().numpy().reshape(-1, 3).view(np.int64))
        x = torch.randn(nx, dtype=torch.float64, device="cuda")
        ys = []
        for var in (3, 7):  # bit 2 off: f64 values + int32 columns; on: codes
            Mat.set_levels(var)
            y = torch.zeros(ny, dtype=torch.float64, device="cuda")
            Mat.mult(v2s, C.c_void_p(x.data_ptr()), None, 0.5, C.c_void_p(y.data_ptr()))
            ys.append(y.cpu().numpy())
        Mat.set_levels(None)
        np.testing.assert_array_equal(ys[0].view(np.int64), ys[1].view(np.int64))
        assert np.abs(ys[0]).max() > 0
    S6, _, _ = make_hip_problem(3, 6, u_deg=2, low_memory=False)
    assert S6._p_vdxi_Mat.vcode is None and S6._divu_Mat.vcode is None  # no dictionary: f64 path, same results


@pytest.mark.parametrize("dim,N,deg,force", [(2, 256, 1, False), (3, 12, 1, True), (3, 6, 2, True), (2, 12, 2, True)])
def test_pair_slot_stream_decodes_to_the_entries_and_spmv_is_bit_identical(dim, N, deg, force):
    """ox_pair_stream_size / _fill (la.SellMatrix.freeze): every slot (col, a, b) stands for the
    entries (col, vdict[a]) and (col + 1, vdict[b]); in stored order they are the row's entries, zeros
    apart.  The mat-vec from the stream equals the entry-stream one bit for bit, for 1..3 columns and
    through the Krylov epilogues."""
    import ctypes as C

    from oasisx_amd import _lib, fem, mesh as M
    from oasisx_amd.fem import cell_geometry
    from oasisx_amd.la import SellMatrix

    lo, hi = [-1.0] * dim, [1.0] * dim
    mesh = M.create_box(None, [lo, hi], [N] * dim) if dim == 3 else M.create_rectangle(None, [lo, hi], [N] * dim)
    V = fem.FunctionSpace(mesh, deg)
    A = SellMatrix(V.pattern)
    lib = _lib.load()
    geom = V.native.nmesh.geom if getattr(V, "native", None) is not None else cell_geometry(mesh, V.local_cells)
    cells = _lib.ox_cells(mesh.gdim, 0, int(geom.shape[0]), geom.data_ptr())
    adj = V.adj.struct()
    nb, bptr, bsl, bw = V.pattern.bins_args()
    _lib.check(lib.ox_assemble_matrix(0, V.degree, C.byref(cells), _lib.ptr(V.cell_dofs), C.byref(adj), _lib.ptr(V.adj.adj_pos),
                                      V.adj.pw, A.ref(), nb, bptr, bsl, bw, _lib.current_stream()), "ox_assemble_matrix")  # mass: SPD
    A.version += 1
    assert A.freeze(pairs="always" if force else "auto")
    assert A.ps_code is not None
    P = V.pattern
    # decode on the host
    ps_ptr = A.ps_ptr.cpu().numpy()
    code = A.ps_code.cpu().numpy().view(np.uint32)
    base = A.ps_base.cpu().numpy().reshape(-1, 2)
    vdict = A.vdict.cpu().numpy()
    ref = P.to_csr(A.vals).tocsr()
    import scipy.sparse as sp
    rows, cols, vals = [], [], []
    for s in range(P.n_slices):
        if ps_ptr[s] & 1:
            continue  # kept in the entry stream
        b0 = ps_ptr[s] & ~255
        ng = ((ps_ptr[s + 1] & ~255) - b0) // 256
        blk = code[b0:b0 + ng * 256].reshape(ng, 64, 4)
        for g in range(ng):
            bb = base[b0 // 256 + g]
            for lane in range(64):
                r = s * 64 + lane
                if r >= P.n_rows:
                    continue
                for j in range(4):
                    cj = int(blk[g, lane, j])
                    col = int(bb[1] if cj & 0x8000 else bb[0]) + (cj & 0x7fff)
                    rows += [r, r]
                    cols += [col, col + 1]
                    vals += [vdict[(cj >> 16) & 0xff], vdict[cj >> 24]]
    wide = [s for s in range(P.n_slices) if ps_ptr[s] & 1]
    dec = sp.coo_matrix((vals, (rows, cols)), shape=(P.n_rows, P.n_cols)).tocsr()
    keep = np.ones(P.n_rows, dtype=bool)
    for s in wide:
        keep[s * 64:(s + 1) * 64] = False
    d = (dec - ref)[keep]
    assert abs(d).max() == 0.0 if d.nnz else True
    for nc in (1, 2, 3):
        x = torch.randn(P.n_cols, nc, dtype=torch.float64, device="cuda").contiguous()
        ys = []
        for var in (7, 15):
            A.set_levels(var)
            y = torch.zeros(P.n_rows, nc, dtype=torch.float64, device="cuda")
            A.mult(x, y, nc)
            ys.append(y)
        A.set_levels(None)
        assert torch.equal(ys[0], ys[1])
    # Krylov epilogues (CG: p.Ap; BiCGStab: D^-1 A with rhat.v and t.t, t.s): same iterates bit for bit
    from oasisx_amd.fem import FieldStorage
    from oasisx_amd.ksp import KSPSolver

    for nc in (1, 3):
        rhs = torch.randn(P.n_rows, nc, dtype=torch.float64, device="cuda")
        for kind in ("cg", "bcgs"):
            sols = []
            for var in (7, 15):
                A.set_levels(var)
                ks = KSPSolver(None, {"ksp_type": kind, "pc_type": "jacobi", "ksp_rtol": 1e-10, "ksp_atol": 1e-30,
                                      "ksp_max_it": 25})
                ks.setOperators(A)
                B, X = FieldStorage(P.n_rows, nc, "cuda"), FieldStorage(P.n_rows, nc, "cuda")
                B.dev().copy_(rhs)
                ks.solve_block(B, X)
                sols.append((X.dev().clone(), ks.iterations))
            A.set_levels(None)
            assert sols[0][1] == sols[1][1] and torch.equal(sols[0][0], sols[1][0]), (kind, nc)
