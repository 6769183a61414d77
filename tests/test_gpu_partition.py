"""GPU: the device kernels on rank-local (owned + ghost) data.  One process builds the local
problem of every rank of a 2- and 3-way partition in turn and checks that the owned rows of
every assembled operator / vector equal the corresponding rows of the unpartitioned problem.
(The collectives themselves -- RCCL halo exchange and all-reduce -- need one GPU per rank and are
exercised by the driver's multi-GPU run; their host logic is covered over gloo in test_dist_cpu.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class FakeComm:
    def __init__(self, rank, size):
        self.rank, self.size, self.handle = rank, size, None

    def allreduce(self, v, op=None):
        return v


def _key(c):
    q = np.round(c * 4096).astype(np.int64)
    k = q[:, 0]
    for j in range(1, q.shape[1]):
        k = k * (1 << 20) + q[:, j]
    return k


def _match(x_global, x_local):
    kg = _key(x_global)
    og = np.argsort(kg)
    return og[np.searchsorted(kg[og], _key(x_local))]


def _problem(dim, N, deg, comm):
    import oasisx_amd as ox
    from oasisx_amd import mesh as M
    from oracle import ipcs_oracle as O
    from tests.helpers import KRYLOV, on_boundary, on_boundary3

    mesh = (M.create_rectangle(comm, [[-1.0, -1.0], [1.0, 1.0]], [N, N]) if dim == 2
            else M.create_box(comm, [[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]], [N, N, N]))
    fns = [O.tg_u, O.tg_v, O.tg_w][:dim]
    marker = on_boundary if dim == 2 else on_boundary3
    bcs = [[ox.DirichletBC(lambda x, f=f: f(x, 0.1, 0.01), ox.LocatorMethod.GEOMETRICAL, marker)] for f in fns]
    S = ox.FractionalStep_AB_CN(mesh, ("Lagrange", deg), ("Lagrange", 1), bcs_u=bcs, bcs_p=[],
                                solver_options=KRYLOV, options={"sell_window": 128}, body_force=(0.1, -0.2, 0.3)[:dim])
    for i in range(dim):
        S._u2[i].interpolate(lambda x, i=i: np.sin(x[0] + i) * np.cos(2 * x[1]) + 0.1 * x[dim - 1])
        S._u1[i].interpolate(lambda x, i=i: np.cos(x[0] - i) * np.sin(x[1]) - 0.2 * x[dim - 1] ** 2)
        S._u[i].interpolate(lambda x, i=i: np.cos(2 * x[0] + i) + x[1] * x[dim - 1])
    S._ps.interpolate(lambda x: np.sin(x[0]) * x[1] + x[dim - 1])
    S._dp.interpolate(lambda x: np.cos(x[0] * x[1]) - x[dim - 1] ** 2)
    return S


@pytest.mark.parametrize("dim,N,deg,P", [(2, 8, 2, 2), (3, 4, 2, 2), (3, 4, 2, 3), (3, 5, 1, 2)])
def test_owned_rows_of_partitioned_operators_match_global(hip, dim, N, deg, P):
    import torch

    dt, nu = 0.05, 0.3
    G = _problem(dim, N, deg, None)
    G.assemble_first(dt, nu)
    G.velocity_tentative_assemble()
    G.pressure_assemble(dt)
    Ag, Mg = G._A.to_scipy(), G._M.to_scipy()
    xg_u, xg_q = G._Vi[0][0].x.cpu().numpy(), G._Q.x.cpu().numpy()
    bfg = G._BFIRST.dev().cpu().numpy()
    rhsg = G._RHS1.dev().cpu().numpy()
    b2g = G._B2.dev().cpu().numpy()[:, 0]
    yg = torch.zeros_like(G._U.dev())
    G._M.mult(G._U.dev(), yg, dim)
    yg = yg.cpu().numpy()
    owned_total = 0
    for r in range(P):
        S = _problem(dim, N, deg, FakeComm(r, P))
        Vi, Q = S._Vi[0][0], S._Q
        assert Vi.n_local > Vi.n_owned > 0
        owned_total += Vi.n_owned
        S.assemble_first(dt, nu)
        S.velocity_tentative_assemble()
        S.pressure_assemble(dt)
        xu = Vi.x.cpu().numpy()
        iu = _match(xg_u, xu)  # global index of every local dof (owned and ghost)
        iq = _match(xg_q, Q.x.cpu().numpy())
        no, nqo = Vi.n_owned, Q.n_owned
        # matrices: owned rows, columns mapped through the coordinates
        for mine, ref in ((S._A.to_scipy(), Ag), (S._M.to_scipy(), Mg)):
            loc = mine.tocoo()
            refd = np.asarray(ref[iu[loc.row], iu[loc.col]]).ravel()
            assert np.abs(loc.data - refd).max() <= 1e-12 * abs(ref).max()
            assert mine.nnz == ref[iu[:no]].nnz
        assert np.abs(S._BFIRST.dev().cpu().numpy()[:no] - bfg[iu[:no]]).max() <= 1e-11 * np.abs(bfg).max()
        assert np.abs(S._RHS1.dev().cpu().numpy()[:no] - rhsg[iu[:no]]).max() <= 1e-11 * np.abs(rhsg).max()
        assert np.abs(S._B2.dev().cpu().numpy()[:nqo, 0] - b2g[iq[:nqo]]).max() <= 1e-11 * np.abs(b2g).max()
        y = torch.zeros_like(S._U.dev())
        S._M.mult(S._U.dev(), y, dim)
        assert np.abs(y.cpu().numpy()[:no] - yg[iu[:no]]).max() <= 1e-12 * np.abs(yg).max()
        # int 1 dx over the owned pressure weights adds up to the volume
        assert S._wQ.shape[0] == nqo
    assert owned_total == G._Vi[0][0].num_dofs


def test_rccl_single_rank_communicator(hip):
    """ncclCommInitRank / halo plan / all-reduce entry points load and run with one rank."""
    import ctypes as C

    import torch

    from oasisx_amd import _lib

    buf = C.create_string_buffer(128)
    _lib.check(hip.ox_comm_unique_id(buf), "ox_comm_unique_id")
    comm = C.c_void_p()
    _lib.check(hip.ox_comm_create(buf.raw, 0, 1, C.byref(comm)), "ox_comm_create")
    d = C.c_void_p()
    z64 = (C.c_int64 * 1)(0)
    _lib.check(hip.ox_dist_create(comm, 0, 1, 0, None, z64, None, z64, 10, 0, C.byref(d)), "ox_dist_create")
    x = torch.arange(10, dtype=torch.float64, device="cuda")
    _lib.check(hip.ox_halo_forward(d, _lib.ptr(x), 1, _lib.current_stream()), "ox_halo_forward")
    _lib.check(hip.ox_allreduce_sum(d, _lib.ptr(x), 10, _lib.current_stream()), "ox_allreduce_sum")
    out = (C.c_double * 4)()
    _lib.check(hip.ox_dot(10, 1, _lib.ptr(x), _lib.ptr(x), out, d, _lib.current_stream()), "ox_dot")
    assert out[0] == 285.0
    _lib.check(hip.ox_dist_destroy(d), "ox_dist_destroy")
    _lib.check(hip.ox_comm_destroy(comm), "ox_comm_destroy")


def test_rccl_calls_of_the_halo_plan_run_on_a_self_loop(hip):
    """The RCCL branch of ox_halo_forward / ox_allreduce_sum -- pack kernel, grouped
    ncclSend/ncclRecv into the ghost block, ncclAllReduce on the caller's stream -- executed for
    real on ONE GPU: a one-rank communicator whose halo plan lists rank 0 itself as the neighbour
    (RCCL matches a send to self with the receive of the same group).  The plan claims two ranks so
    that no single-rank shortcut is taken."""
    import ctypes as C

    import numpy as np
    import torch

    from oasisx_amd import _lib

    buf = C.create_string_buffer(128)
    _lib.check(hip.ox_comm_unique_id(buf), "ox_comm_unique_id")
    comm = C.c_void_p()
    _lib.check(hip.ox_comm_create(buf.raw, 0, 1, C.byref(comm)), "ox_comm_create")
    n_owned, send = 12, [3, 0, 7, 11, 5]
    ng = len(send)
    peers = np.asarray([0], dtype=np.int32)
    off = np.asarray([0, ng], dtype=np.int64)
    send_idx = torch.tensor(send, dtype=torch.int32, device="cuda")
    d = C.c_void_p()
    _lib.check(hip.ox_dist_create(comm, 0, 2, 1, peers.ctypes.data_as(C.POINTER(C.c_int32)),
                                  off.ctypes.data_as(C.POINTER(C.c_int64)), _lib.ptr(send_idx),
                                  off.ctypes.data_as(C.POINTER(C.c_int64)), n_owned, ng, C.byref(d)), "ox_dist_create")
    for nc in (1, 3):
        x = torch.zeros(n_owned + ng, nc, dtype=torch.float64, device="cuda")
        x[:n_owned] = torch.arange(n_owned * nc, dtype=torch.float64, device="cuda").reshape(n_owned, nc) + 1.0
        x[n_owned:] = float("nan")
        _lib.check(hip.ox_halo_forward(d, _lib.ptr(x), nc, _lib.current_stream()), "ox_halo_forward")
        torch.cuda.synchronize()
        assert torch.equal(x[n_owned:], x[torch.tensor(send, device="cuda")])
    s = torch.tensor([1.5, -2.0, 3.25], dtype=torch.float64, device="cuda")
    _lib.check(hip.ox_allreduce_sum(d, _lib.ptr(s), 3, _lib.current_stream()), "ox_allreduce_sum")
    torch.cuda.synchronize()
    assert s.tolist() == [1.5, -2.0, 3.25]  # sum over the communicator's one rank
    _lib.check(hip.ox_dist_destroy(d), "ox_dist_destroy")
    _lib.check(hip.ox_comm_destroy(comm), "ox_comm_destroy")


def test_overlapped_spmv_on_the_rccl_self_loop(hip):
    """The overlapped distributed mat-vec (ox_sell.ib_slices: halo exchange started on the side stream,
    interior slices multiplied meanwhile, boundary slices after the exchange) over the RCCL branch,
    on the one-rank self-loop plan: interior and boundary launches, their partial sums and the
    event hand-over between the two streams are all exercised for real."""
    import ctypes as C

    import numpy as np
    import torch

    from oasisx_amd import _lib, fem
    from oasisx_amd.la import SellMatrix

    buf = C.create_string_buffer(128)
    _lib.check(hip.ox_comm_unique_id(buf), "ox_comm_unique_id")
    comm = C.c_void_p()
    _lib.check(hip.ox_comm_create(buf.raw, 0, 1, C.byref(comm)), "ox_comm_create")
    n_owned, send = 192, [3, 0, 7, 150, 64]
    ng = len(send)
    peers = np.asarray([0], dtype=np.int32)
    off = np.asarray([0, ng], dtype=np.int64)
    send_idx = torch.tensor(send, dtype=torch.int32, device="cuda")
    d = C.c_void_p()
    _lib.check(hip.ox_dist_create(comm, 0, 2, 1, peers.ctypes.data_as(C.POINTER(C.c_int32)),
                                  off.ctypes.data_as(C.POINTER(C.c_int64)), _lib.ptr(send_idx),
                                  off.ctypes.data_as(C.POINTER(C.c_int64)), n_owned, ng, C.byref(d)), "ox_dist_create")
    # rows 0..191 (3 slices); slices 1 and 2 reference ghost columns, slice 0 does not
    n_cols = n_owned + ng
    rows, cols = [], []
    for r in range(n_owned):
        cs = {r, (r + 1) % n_owned, (r * 7 + 3) % n_owned}
        if r == 5:
            cs.add(n_owned - 1)  # an INTERIOR row next to the ghost block: a pair slot must not read x[n_owned]
        if r >= 64:
            cs.add(n_owned + r % ng)
        for c in sorted(cs):
            rows.append(r)
            cols.append(c)
    keys = torch.tensor([r * n_cols + c for r, c in zip(rows, cols)], dtype=torch.int64, device="cuda")
    rl = torch.bincount(torch.tensor(rows, device="cuda"), minlength=n_owned)
    rp = torch.zeros(n_owned + 1, dtype=torch.int64, device="cuda")
    rp[1:] = torch.cumsum(rl, 0)
    P = fem.build_sell(n_owned, n_cols, keys, rl, rp)
    P.dist = d
    P.split_interior(n_owned)
    assert P.n_interior == 1 and P.ib_slices.tolist() == [0, 1, 2]
    A = SellMatrix(P)
    g = torch.Generator(device="cuda").manual_seed(2)
    A.vals.copy_(torch.rand(P.size, dtype=torch.float64, device="cuda", generator=g))
    csr = A.to_scipy()  # the real entries
    A.vals.copy_(P.values_from_csr(csr))  # padding slots hold 0, as every assembled matrix has them
    dense = torch.from_numpy(csr.toarray()).cuda()
    for nc in (1, 3):
        x = torch.zeros(n_cols, nc, dtype=torch.float64, device="cuda")
        x[:n_owned] = torch.rand(n_owned, nc, dtype=torch.float64, device="cuda", generator=g)
        x[n_owned:] = float("nan")  # must be overwritten by the exchange before the boundary slices read it
        y = torch.zeros(n_owned, nc, dtype=torch.float64, device="cuda")
        for _ in range(3):  # repeated: the event pair is reused
            A.mult(x, y, nc)
        torch.cuda.synchronize()
        xf = x.clone()
        xf[n_owned:] = x[torch.tensor(send, device="cuda")]
        assert torch.equal(x[n_owned:], xf[n_owned:])
        assert (y - dense @ xf).abs().max() < 1e-13
    # the same through the pair-slot stream of a frozen matrix (interior / boundary slice lists included)
    pal = torch.tensor([0.25, 0.5, 0.75, 1.0, 1.5, 2.0], dtype=torch.float64, device="cuda")
    csr.data[:] = pal[torch.randint(0, 6, (csr.nnz,), device="cuda", generator=g)].cpu().numpy()
    A.vals.copy_(P.values_from_csr(csr))
    A.version += 1
    assert A.freeze(pairs="always") and A.ps_code is not None
    # invariant of the stream: a slot reads x[col] and x[col+1]; in an INTERIOR slice (multiplied while the
    # exchange is in flight) neither may be a ghost column, not even with a zero coefficient
    ps_ptr, code = A.ps_ptr.cpu().numpy(), A.ps_code.cpu().numpy().view(np.uint32)
    base = A.ps_base.cpu().numpy().reshape(-1, 2)
    for sl in P.ib_slices[: P.n_interior].tolist():
        b0 = int(ps_ptr[sl]) & ~255
        ng_ = ((int(ps_ptr[sl + 1]) & ~255) - b0) // 256
        blk = code[b0:b0 + ng_ * 256].reshape(ng_, 64, 4)
        for g_ in range(ng_):
            lo_hi = base[b0 // 256 + g_]
            cols_ = np.where(blk[g_] & 0x8000, lo_hi[1], lo_hi[0]) + (blk[g_] & 0x7fff)
            assert int(cols_.max()) + 1 < n_owned, (sl, g_, int(cols_.max()))
    dense = torch.from_numpy(csr.toarray()).cuda()
    for nc in (1, 3):
        x = torch.zeros(n_cols, nc, dtype=torch.float64, device="cuda")
        x[:n_owned] = torch.rand(n_owned, nc, dtype=torch.float64, device="cuda", generator=g)
        ys = []
        for var in (7, 15):
            A.set_levels(var)
            x[n_owned:] = float("nan")
            y = torch.zeros(n_owned, nc, dtype=torch.float64, device="cuda")
            A.mult(x, y, nc)
            torch.cuda.synchronize()
            ys.append(y)
        A.set_levels(None)
        assert torch.equal(ys[0], ys[1])
        assert (ys[1] - dense @ x).abs().max() < 1e-13
    _lib.check(hip.ox_dist_destroy(d), "ox_dist_destroy")
    _lib.check(hip.ox_comm_destroy(comm), "ox_comm_destroy")
