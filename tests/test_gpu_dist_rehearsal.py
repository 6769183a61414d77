"""GPU, 2 (and 3) processes on ONE device: the whole mesh-partitioned step -- partitioned assembly,
the C Krylov loops with their halo exchanges and merged all-reduces, ghost-consistency of every
field -- against the serial run.  RCCL refuses several ranks on one GPU, so the exchange points go
through (a) the direct xGMI transport -- the ranks' uncached windows mapped into each other through
HIP IPC, exactly as between GPUs, the "remote" stores landing on the same device -- and (b) the
library's callback transport (gloo, host staged); everything else (call sites, kernels, host
logic) is the production path."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _key(c):
    q = np.round(c * 4096).astype(np.int64)
    k = q[:, 0]
    for j in range(1, q.shape[1]):
        k = k * (1 << 20) + q[:, j]
    return k


def _run(dim, N, deg, comm, steps, low_memory=True, p_deg=1):
    import oasisx_amd as ox
    from oasisx_amd import mesh as M
    from oracle import ipcs_oracle as O
    from tests.helpers import KRYLOV, on_boundary, on_boundary3

    nu, dt = 0.01, 0.005
    if N < 0:  # an UNSTRUCTURED mesh (Delaunay, refined once): Z-order numbering, the mat-vecs on the LDS-window stream
        mesh = M.create_delaunay_box(comm, [[-1.0] * dim, [1.0] * dim], -N, refine=1)
    else:
        mesh = (M.create_rectangle(comm, [[-1.0, -1.0], [1.0, 1.0]], [N, N]) if dim == 2
                else M.create_box(comm, [[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]], [N, N, N]))
    clock = {"t": 0.0}
    fns = [O.tg_u, O.tg_v, O.tg_w][:dim]
    marker = on_boundary if dim == 2 else on_boundary3
    bcs = [[ox.DirichletBC(lambda x, f=f: f(x, clock["t"], nu), ox.LocatorMethod.GEOMETRICAL, marker)] for f in fns]
    opts = {k: dict(v, ksp_initial_guess_nonzero=True) for k, v in KRYLOV.items()}
    S = ox.FractionalStep_AB_CN(mesh, ("Lagrange", deg), ("Lagrange", p_deg), bcs_u=bcs, bcs_p=[], solver_options=opts,
                                options={"sell_window": 128, "low_memory_version": low_memory})
    for i, f in enumerate(fns):
        S._u2[i].interpolate(lambda x, f=f: f(x, -dt, nu))
        S._u1[i].interpolate(lambda x, f=f: f(x, 0.0, nu))
    S._p.interpolate(lambda x: O.tg_p(x, -dt / 2, nu))
    diffs = []
    for _ in range(steps):
        clock["t"] += dt
        diffs.append(S.solve(dt, nu, max_iter=1))
    return S, diffs


def _worker(rank, world, port, dim, N, deg, low_memory, transport, out, release=None):
    import torch.distributed as dist

    if release is not None:
        os.environ["OX_P2P_RELEASE"] = release
    os.environ["OX_TRANSPORT"] = transport
    os.environ["OX_P2P_TIMEOUT_S"] = "30"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oasisx_amd.parallel import init_comm

        comm = init_comm()
        assert comm.size == world and comm.handle is None
        p_deg = 2 if deg == 3 else 1  # (P3-P2 Taylor-Hood, round 5: face / cell dofs have owners and halo entries)
        S, diffs = _run(dim, N, deg, comm, steps=2, low_memory=low_memory, p_deg=p_deg)
        G, gdiffs = _run(dim, N, deg, None, steps=2, low_memory=low_memory, p_deg=p_deg)
        Vi, Q = S._Vi[0][0], S._Q
        assert Vi.dist is not None and Vi.n_local > Vi.n_owned
        assert comm.active == {deg: transport, p_deg: transport}, comm.active
        assert Vi.num_dofs_global == G._Vi[0][0].num_dofs and Q.num_dofs_global == G._Q.num_dofs
        if N < 0:  # the partitioned operators really run on window blocks, split interior / boundary
            for Pn in (S._A.pattern, S._Ap.pattern):
                assert Pn.wcode is not None and Pn.n_wb_interior is not None and 0 <= Pn.n_wb_interior <= Pn.n_wblocks
            assert S._A._struct.n_wblocks == S._A.pattern.n_wblocks and S._A._struct.n_wb_interior == S._A.pattern.n_wb_interior
        kg = _key(G._Vi[0][0].x.cpu().numpy())
        og = np.argsort(kg)
        iu = og[np.searchsorted(kg[og], _key(Vi.x.cpu().numpy()))]
        kq = _key(G._Q.x.cpu().numpy())
        oq = np.argsort(kq)
        iq = oq[np.searchsorted(kq[oq], _key(Q.x.cpu().numpy()))]
        ug = G._U1.dev().cpu().numpy()
        pg = G._P.dev().cpu().numpy()[:, 0]
        ul = S._U1.dev().cpu().numpy()
        pl = S._P.dev().cpu().numpy()[:, 0]
        # owned AND ghost entries agree with the serial run (ghosts are kept consistent)
        du = float(np.abs(ul - ug[iu]).max())
        dp = float(np.abs(pl - pg[iq]).max())
        assert du < 1e-8 and dp < 1e-7, (du, dp)
        assert abs(diffs[-1] - gdiffs[-1]) < 1e-8 * max(1.0, abs(gdiffs[-1]))
        assert abs(S._vol - G._vol) < 1e-12
        out[rank] = (du, dp, S.iteration_counts(), G.iteration_counts())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("transport", ["p2p", "host"])
@pytest.mark.parametrize("dim,N,deg,world,low_memory", [(3, 6, 2, 2, True), (2, 12, 2, 3, True), (3, 6, 1, 2, True),
                                                        (3, 6, 2, 2, False), (3, -5, 2, 2, True), (2, -14, 2, 3, False),
                                                        (3, 4, 3, 2, True), (3, 3, 3, 3, False), (2, 10, 3, 2, True), (2, 9, 3, 3, False)])
def test_partitioned_steps_match_serial(hip, dim, N, deg, world, low_memory, transport):
    import torch.multiprocessing as mp


    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), dim, N, deg, low_memory, transport, out), nprocs=world, join=True)
    assert len(out) == world, dict(out)


@pytest.mark.parametrize("dim,N,deg,world", [(3, 6, 2, 2), (2, 12, 2, 3)])
def test_partitioned_steps_with_the_fast_release_protocol(hip, dim, N, deg, world):
    """The window transport's default is the CONSERVATIVE release protocol (per-wave system fences, release-scope flag
    stores: ox_dist_set_p2p_release); OX_P2P_RELEASE=fast selects round 5's one-fence form.  Both must give the serial
    run's fields -- here with every rank on one device, the only place either has run."""
    import torch.multiprocessing as mp

    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), dim, N, deg, True, "p2p", out, "fast"), nprocs=world, join=True)
    assert len(out) == world, dict(out)


def _retest_worker(rank, world, port, transport, out):
    """Merged-reduction BiCGStab (the DEFAULT on a partitioned operator) on D^-1 A = I + 1e-9 E at rtol 1e-25: its
    recurrence norm is rounding noise after one iteration (tests/test_gpu_ksp_options.py); the stored residual decides,
    through an all-reduced r.r, identically on every rank."""
    import torch.distributed as dist

    os.environ["OX_TRANSPORT"] = transport
    os.environ["OX_P2P_TIMEOUT_S"] = "30"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oasisx_amd import _lib
        from oasisx_amd.fem import FieldStorage
        from oasisx_amd.ksp import KSPSolver
        from oasisx_amd.la import SellMatrix
        from oasisx_amd.parallel import init_comm

        comm = init_comm()
        res = {}
        for tag, cm in (("part", comm), ("serial", None)):
            S, _ = _run(2, 12, 2, cm, steps=0)
            Vi = S._Vi[0][0]
            P = S._K.pattern
            assert (P.dist is not None) == (cm is not None)
            rows_, k_ = P.slot_rows_k()
            rl = np.zeros(P.n_slices * 64, dtype=np.int64)
            rl[: P.n_rows] = P.row_len.cpu().numpy()
            real = k_ < rl[rows_]
            diag = torch.from_numpy((P.cols.cpu().numpy() == rows_) & real).cuda()
            x = Vi.x.cpu().numpy()
            A = SellMatrix(P, name="A")
            A.vals.copy_(1e-9 * S._K.vals)
            # the diagonal and the right-hand sides as functions of the dof's position: the same system on every layout
            d = 1.5 + 0.5 * np.sin(11.0 * x[:, 0] + 3.0) * np.cos(7.0 * x[:, 1])
            A.vals[diag] += torch.from_numpy(d[rows_[diag.cpu().numpy()]]).cuda()
            A.version += 1
            nl, no = Vi.n_local, Vi.n_owned
            cols = [np.sin(37.0 * x[:, 0]) * np.cos(23.0 * x[:, 1]) + 0.3, np.cos(5.0 * x[:, 0] * x[:, 1]) + x[:, 1],
                    np.sin(91.0 * x[:, 0] + 17.0 * x[:, 1])]
            B = FieldStorage(nl, 3, "cuda")
            B.dev()[:] = torch.from_numpy(np.stack(cols, axis=1)).cuda()
            ksp = KSPSolver(cm, {"ksp_type": "bcgs", "pc_type": "jacobi", "ksp_rtol": 1e-25, "ksp_atol": 1e-300, "ksp_max_it": 50})
            ksp.setOperators(A)
            assert ksp._method()[0] == (_lib.KSP_BCGS_MERGED if cm is not None else _lib.KSP_BCGS)
            X = FieldStorage(nl, 3, "cuda")
            reasons = ksp.solve_block(B, X)
            r = ksp.last_result
            res[tag] = (reasons, list(r.its[:3]), list(r.rnorm[:3]), list(r.bnorm[:3]), list(r.resumed[:3]),
                        X.dev().cpu().numpy().copy(), x, no)
        (rp, ip, rnp, bnp, resp, xp, cp, no), (rs, is_, rns, bns, _, xs, cs, _) = res["part"], res["serial"]
        assert rp == [2, 2, 2] and rs == [2, 2, 2], (rp, rs)
        for c in range(3):
            assert rnp[c] <= 1e-25 * bnp[c], (c, rnp, bnp, resp)
            assert abs(ip[c] - is_[c]) <= 1
            assert abs(bnp[c] - bns[c]) <= 1e-12 * bns[c]  # the all-reduced norm of b is the serial one
        kg = _key(cs)
        og = np.argsort(kg)
        iu = og[np.searchsorted(kg[og], _key(cp))]
        assert np.abs(xp - xs[iu]).max() <= 1e-14 * np.abs(xs).max()  # owned AND ghost entries (scatter_forward)
        out[rank] = (tuple(resp), tuple(ip))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("transport", ["p2p", "host"])
def test_partitioned_merged_bicgstab_retests_the_stored_residual(hip, transport):
    import torch.multiprocessing as mp

    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_retest_worker, args=(2, _free_port(), transport, out), nprocs=2, join=True)
    assert len(out) == 2 and out[0] == out[1], dict(out)  # the same re-openings, the same iteration counts on both ranks
    assert sum(out[0][0]) > 0, dict(out)  # at least one column was re-opened


def test_first_link_check_script_rehearsal(hip):
    """tools/first_link_check.py -- the procedure for the first node with a link (halo self-test, bit-exact all-reduce,
    two steps against the serial run, 200 timed exchanges, for every device transport) -- stays runnable: its own
    rehearsal mode, 2 ranks sharing this GPU over gloo, conservative and fast release protocol and the host transport."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "OX_TRANSPORT", "OX_P2P_RELEASE")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "tools", "first_link_check.py"), "--backend", "gloo", "-N", "6"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    recs = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert [x["transport"] for x in recs[:-1]] == ["p2p", "p2p-fast", "host"]
    for x in recs[:-1]:
        assert x["max_abs_du"] < 1e-8 and x["allreduce_bits_equal_on_all_ranks"] and x["allreduce_is_the_rank_order_sum"]
        assert set(x["active"].values()) == {x["transport"].split("-")[0]}
    assert recs[-1]["first_link_check"] == {"p2p": "passed", "p2p-fast": "passed", "host": "passed"}
    assert "no link was crossed" in recs[-1]["note"]
