"""GPU: the LDS-window stream of a SELL-64 pattern (``ox_sell.wb_*`` / ``wlist`` / ``wt_ptr`` / ``wcode`` / ``wvcode``,
kernel ``k_spmv_win``): an optional storage level of the velocity matrices' mat-vecs (reference fracstep.py:452,521,
615,634 -> ``Mat.mult`` / ``KSP.solve``) that reads the x operands from an LDS copy of a block's window instead of one
gather per entry.  Same entries, same per-row order of fused multiply-adds: bit-identical y on every mesh; whole time
steps with ``options["spmv_windows"]`` (brick order of the numbering + windows) agree with the oracle like the default."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _space(kind, dim, n, brick, setup="native"):
    import os

    from oasisx_amd import fem
    from oasisx_amd import mesh as M

    old = os.environ.get("OX_SETUP")
    os.environ["OX_SETUP"] = setup  # "torch": fem.py's twin of the library set-up (and of its window builder)
    try:
        return _space_impl(kind, dim, n, brick, fem, M)
    finally:
        if old is None:
            del os.environ["OX_SETUP"]
        else:
            os.environ["OX_SETUP"] = old


def _space_impl(kind, dim, n, brick, fem, M):

    if kind == "box":
        mesh = (M.create_rectangle(None, [[-1.0, -1.0], [1.0, 1.0]], [n, n]) if dim == 2 else
                M.create_box(None, [[-1.0] * 3, [1.0] * 3], [n, n + 1, n - 1]))
    else:
        mesh = M.create_delaunay_box(None, [[-1.0] * dim, [1.0] * dim], n, refine=1)
    return fem.FunctionSpace(mesh, 2, window=1024, brick=brick)


@pytest.mark.parametrize("setup,split", [("native", 0), ("torch", 0), ("native", 1200)])
@pytest.mark.parametrize("kind,dim,n,brick", [("box", 3, 17, True), ("box", 3, 9, False), ("box", 2, 40, True),
                                             ("delaunay", 3, 6, False), ("delaunay", 2, 14, False)])
def test_window_stream_reproduces_the_columns_and_the_matvec_bit_for_bit(hip, kind, dim, n, brick, setup, split):
    """``split``: blocks whose window holds more entries are cut in two (``ox_space_windows_split``, what the solver asks
    for on its velocity pattern): the same checks on blocks of 4 + 4 slices."""
    from oasisx_amd import _lib
    from oasisx_amd.la import SellMatrix

    V = _space(kind, dim, n, brick, setup)
    assert (V.native is not None) == (setup == "native")
    assert V.build_windows(split)
    P = V.pattern
    if split:
        wsz = (P.wb_ptr[1:] - P.wb_ptr[:-1]).cpu().numpy()
        cnt = (P.wb_slices.cpu().numpy() >= 0).sum(axis=1)
        assert ((wsz <= split) | (cnt <= 4)).all()  # a block over the bound has been cut (one level: 4 slices may stay over)
        if dim == 3:
            assert (cnt <= 4).any(), "the bound is meant to cut blocks of these meshes"
    # ---- structure: every slice in exactly one block, every slot's column recovered from its window ----------------
    sl = P.wb_slices.cpu().numpy()
    used = np.sort(sl[sl >= 0])
    assert (used == np.arange(P.n_slices)).all()
    wb_ptr, wlist = P.wb_ptr.cpu().numpy(), P.wlist.cpu().numpy()
    wt_ptr = P.wt_ptr.cpu().numpy()
    wcode = P.wcode.cpu().numpy().view(np.uint16)
    cols, sp = P.cols.cpu().numpy(), P.slice_ptr.cpu().numpy()
    sched = P.wb_waves.cpu().numpy().view(np.uint16)
    for b in range(sl.shape[0]):
        win = wlist[wb_ptr[b]:wb_ptr[b + 1]]
        assert (np.diff(win) > 0).all()  # ascending, distinct
        loads = np.zeros(4, dtype=np.int64)
        for j, s in enumerate(sl[b]):
            if s < 0:
                continue
            npair = (sp[s + 1] - sp[s]) // 128
            loads[(sched[b] >> (2 * j)) & 3] += npair
            c = cols[sp[s]:sp[s + 1]].reshape(npair, 64, 2)  # [pair][lane][2]
            nt = (npair + 1) // 2
            t = wcode[wt_ptr[s] * 256:(wt_ptr[s] + nt) * 256].reshape(nt, 64, 2, 2)  # [tile][lane][pair in tile][2]
            dec = win[t.astype(np.int64)].transpose(0, 2, 1, 3).reshape(nt * 2, 64, 2)[:npair]
            assert (dec == c).all()
        tot = loads.sum()
        assert loads.max() <= max(tot / 4 * 1.5, (sp[sl[b][sl[b] >= 0] + 1] - sp[sl[b][sl[b] >= 0]]).max() // 128)  # balanced
    # ---- mat-vec: window level against the lane = row level, f64 values and value codes, 1..3 right-hand sides --------
    lib = _lib.load()
    rows_, k_ = P.slot_rows_k()
    rl = np.zeros(P.n_slices * 64, dtype=np.int64)
    rl[: P.n_rows] = P.row_len.cpu().numpy()
    real = torch.from_numpy(k_ < rl[rows_]).cuda()  # padding slots carry the value 0 (and the row's own column)
    A = SellMatrix(P, name="A")
    A.vals.copy_((torch.rand(P.size, dtype=torch.float64, device="cuda") - 0.3) * real)
    Md = SellMatrix(P, name="D")
    pal = torch.tensor([0.0, 1.5, -2.25, 1e-3, 7.0, -0.125, 3.0, 0.5], dtype=torch.float64, device="cuda")
    Md.vals.copy_(pal[torch.randint(0, 8, (P.size,), device="cuda")] * real)
    Md.version += 1
    assert Md.freeze(pairs="never") and Md.wvcode is not None
    try:
        for nc in (1, 2, 3):
            x = torch.randn(P.n_cols, nc, dtype=torch.float64, device="cuda")
            for Mat in (A, Md):
                y0, y1 = torch.zeros(P.n_rows, nc, dtype=torch.float64, device="cuda"), torch.full((P.n_rows, nc), 9.0, dtype=torch.float64, device="cuda")
                Mat.set_levels(15)
                Mat.mult(x, y0, nc)
                Mat.set_levels(31)
                Mat.mult(x, y1, nc)
                assert torch.equal(y0, y1)
                ref = Mat.to_scipy() @ x.cpu().numpy()
                assert np.abs(y1.cpu().numpy() - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
        x = torch.randn(P.n_cols, 3, dtype=torch.float64, device="cuda")
        y0, y1 = torch.zeros(P.n_rows, 3, dtype=torch.float64, device="cuda"), torch.zeros(P.n_rows, 3, dtype=torch.float64, device="cuda")
        A.set_levels(15)
        A.mult(x, y0, 3)
    finally:
        A.set_levels(None)
        Md.set_levels(None)
    assert torch.equal(y0, (A.mult(x, y1, 3), y1)[1])


@pytest.mark.parametrize("dim,N", [(3, 8), (2, 32)])  # h a power of two: bit-identical cells, M carries a value dictionary
def test_krylov_solves_and_time_steps_on_the_window_stream(hip, dim, N):
    """options["spmv_windows"]: every mat-vec of the velocity matrices -- BiCGStab and CG epilogues, lockstep and
    narrowed columns, M through its value dictionary, A with f64 values -- runs on the window stream; the step agrees
    with the oracle exactly as the default storage does (tests/test_gpu_parity.py)."""
    import oasisx_amd as ox
    from oracle import ipcs_oracle as O
    from tests.helpers import KRYLOV, make_oracle_twin, on_boundary, on_boundary3, tg_mesh

    nu, dt = 0.01, 0.005
    mesh = tg_mesh(dim, N)
    clock = {"t": 0.0}
    fns = [O.tg_u, O.tg_v, O.tg_w][:dim]
    marker = on_boundary if dim == 2 else on_boundary3
    S = ox.FractionalStep_AB_CN(
        mesh, ("Lagrange", 2), ("Lagrange", 1), bcs_p=[], solver_options=KRYLOV,
        bcs_u=[[ox.DirichletBC(lambda x, f=f: f(x, clock["t"], nu), ox.LocatorMethod.GEOMETRICAL, marker)] for f in fns],
        options={"sell_window": 1024, "spmv_windows": True, "low_memory_version": False})
    P = S._M.pattern
    assert S._Vi[0][0].brick and P.wcode is not None and S._M.wvcode is not None and S._A._struct.n_wblocks == P.n_wblocks > 0
    for i, f in enumerate(fns):
        S._u2[i].interpolate(lambda x, f=f: f(x, -dt, nu))
        S._u1[i].interpolate(lambda x, f=f: f(x, 0.0, nu))
    S._p.interpolate(lambda x: O.tg_p(x, -dt / 2.0, nu))
    R, rclock = make_oracle_twin(S, mesh, dim, 2, nu, dt, solver_options=KRYLOV)
    t = 0.0
    for _ in range(2):
        t += dt
        clock["t"] = rclock["t"] = t
        S.solve(dt, nu, max_iter=1)
        R.solve(dt, nu, max_iter=1)
    u = S.u.x.array.reshape(-1, dim)
    assert np.abs(u - R.u1).max() < 1e-8 and np.abs(S._p.x.array - R.p).max() < 1e-7
    its, its_o = S.iteration_counts(), R.its
    for k in ("tentative", "update"):
        assert all(abs(a - b) <= 1 for a, b in zip(its[k][:dim], its_o[k])), (its, its_o)


def test_krylov_workspace_covers_the_window_grid_of_a_short_sort_window(hip):
    """with a 64-row length-sort window every slice is its own window block, so the window mat-vec
    launches n_slices blocks -- four times the lane = row grid the Krylov partial-sum arrays used to be sized for.  On an
    operator of more than 4096 slices a three-column BiCGStab then wrote past them into the Krylov vectors.  The workspace
    is sized from the operator's real grid now (``ox_ksp_work_bytes_for``): same solution as on the lane = row kernels."""
    from oasisx_amd import _lib, fem
    from oasisx_amd import mesh as M
    from oasisx_amd.fem import FieldStorage
    from oasisx_amd.ksp import KSPSolver
    from oasisx_amd.la import SellMatrix

    lib = _lib.load()
    mesh = M.create_rectangle(None, [[-1.0, -1.0], [1.0, 1.0]], [256, 256])
    V = fem.FunctionSpace(mesh, 2, window=64)
    assert V.build_windows()
    P = V.pattern
    assert P.n_slices > 4096 and P.n_wblocks > (P.n_slices + 3) // 4 + 64
    A = SellMatrix(P, name="A")
    rows_, k_ = P.slot_rows_k()
    rl = np.zeros(P.n_slices * 64, dtype=np.int64)
    rl[: P.n_rows] = P.row_len.cpu().numpy()
    real = torch.from_numpy(k_ < rl[rows_]).cuda()
    g = torch.Generator(device="cuda").manual_seed(3)
    A.vals.copy_((torch.rand(P.size, dtype=torch.float64, device="cuda", generator=g) - 0.5) * real)
    # diagonally dominant: the diagonal slot of each row (column == row) gets 40
    diag = torch.from_numpy(P.cols.cpu().numpy() == rows_).cuda() & real
    A.vals[diag] = 40.0
    A.version += 1
    assert lib.ox_ksp_work_bytes_for(A.ref(), 3, _lib.KSP_BCGS_MERGED) > lib.ox_ksp_work_bytes(P.n_rows, P.n_cols, 3, _lib.KSP_BCGS_MERGED)
    n = P.n_rows
    B = FieldStorage(n, 3, "cuda")
    B.dev()[:] = torch.randn(n, 3, dtype=torch.float64, device="cuda", generator=g)
    sols = {}
    try:
        for variant in (15, 31):
            A.set_levels(variant)
            for merged in (False, True):
                ksp = KSPSolver(None, {"ksp_type": "bcgs", "pc_type": "jacobi", "ksp_rtol": 1e-10, "ksp_bcgs_merged_reduction": merged})
                ksp.setOperators(A)
                X = FieldStorage(n, 3, "cuda")
                assert ksp.solve_block(B, X) == [2, 2, 2]
                sols[(variant, merged)] = (X.dev().clone(), list(ksp.iterations[:3]))
    finally:
        A.set_levels(None)
    ref = sols[(15, False)]
    y = torch.zeros(n, 3, dtype=torch.float64, device="cuda")
    for key, (x, its) in sols.items():
        assert all(abs(a - b) <= 2 for a, b in zip(its, ref[1])), (key, its, ref[1])
        assert (x - ref[0]).abs().max() < 1e-8 * ref[0].abs().max(), key
        A.mult(x, y, 3)
        assert (y - B.dev()).abs().max() < 1e-7 * B.dev().abs().max(), key
