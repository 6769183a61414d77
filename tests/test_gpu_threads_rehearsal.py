"""GPU, EIGHT ranks of a mesh-partitioned job in ONE process: whole IPCS steps on the 2 x 2 x 2 split an 8-GPU run gets.

A GPU box takes at most six processes on its card, so the 2-4 rank process rehearsals (tests/test_gpu_dist_rehearsal.py)
stop short of the rank count of BASELINE.json's multi-GPU configurations.  Here every rank is a host THREAD driving its
own part -- sub-mesh, partitioned spaces, operators, Krylov solvers -- through the library's callback transport
(``ox_dist_create_custom``): the blocking Krylov loops of the eight ranks run side by side and meet at every halo
exchange and all-reduce (tests/helpers.py ``ThreadWorld``), as the ranks of an RCCL job meet in ncclSend/ncclRecv and
ncclAllReduce (reference: ``scatter_forward`` after every product, fracstep.py:453,497,502,551,632,655; ksp.py:77).
Owned and ghost entries of every rank's fields are compared with the serial run's after two steps."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _q(x):
    return np.round((np.asarray(x, dtype=np.float64) + 1.0) * float(1 << 35)).astype(np.int64)


def _lookup(xg):
    q = _q(xg)
    return {tuple(k): i for i, k in enumerate(q.tolist())}


def _run(N, deg, p_deg, comm, steps, kind="box", low_memory=True):
    import oasisx_amd as ox
    from oasisx_amd import mesh as M
    from oracle import ipcs_oracle as O
    from tests.helpers import KRYLOV, on_boundary3

    nu, dt = 0.01, 0.005
    if kind == "delaunay":
        mesh = M.create_delaunay_box(comm, [[-1.0] * 3, [1.0] * 3], N, seed=4)
    else:
        mesh = M.create_box(comm, [[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]], [N, N, N])
    clock = {"t": 0.0}
    fns = [O.tg_u, O.tg_v, O.tg_w]
    bcs = [[ox.DirichletBC(lambda x, f=f: f(x, clock["t"], nu), ox.LocatorMethod.GEOMETRICAL, on_boundary3)] for f in fns]
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


@pytest.mark.parametrize("N,deg,p_deg,kind,low_memory", [(8, 2, 1, "box", True), (6, 2, 1, "box", False), (8, 1, 1, "box", True),
                                                         (4, 3, 2, "box", True), (6, 2, 1, "delaunay", True)])
def test_eight_rank_threads_step_like_the_serial_run(hip, N, deg, p_deg, kind, low_memory):
    from tests.helpers import run_rank_threads

    G, gdiffs = _run(N, deg, p_deg, None, 2, kind, low_memory)
    torch.cuda.synchronize()
    lu, lq = _lookup(G._Vi[0][0].x.cpu().numpy()), _lookup(G._Q.x.cpu().numpy())
    ug, pg = G._U1.dev().cpu().numpy(), G._P.dev().cpu().numpy()[:, 0]

    def rank_job(comm):
        S, diffs = _run(N, deg, p_deg, comm, 2, kind, low_memory)
        torch.cuda.synchronize()
        Vi, Q = S._Vi[0][0], S._Q
        assert Vi.dist is not None and comm.active == {deg: "host", p_deg: "host"}, comm.active
        iu = np.asarray([lu[tuple(k)] for k in _q(Vi.x.cpu().numpy()).tolist()])
        iq = np.asarray([lq[tuple(k)] for k in _q(Q.x.cpu().numpy()).tolist()])
        ul, pl = S._U1.dev().cpu().numpy(), S._P.dev().cpu().numpy()[:, 0]
        return {"du": float(np.abs(ul - ug[iu]).max()), "dp": float(np.abs(pl - pg[iq]).max()), "diff": diffs[-1],
                "vol": S._vol, "owned": (int(Vi.n_owned), int(Q.n_owned)), "ghosts": int(Vi.n_local - Vi.n_owned),
                "peers": [int(p) for p in Vi.halo["peers"]], "its": {k: [int(i) for i in v] for k, v in S.iteration_counts().items()}}

    res, world = run_rank_threads(8, rank_job)
    assert sum(r["owned"][0] for r in res) == G._Vi[0][0].num_dofs and sum(r["owned"][1] for r in res) == G._Q.num_dofs
    if kind == "box":
        assert len(res[0]["peers"]) == 7 and len(res[7]["peers"]) == 7  # the 2 x 2 x 2 split: the corner octants meet everyone
    for r in res:
        assert r["du"] < 1e-8 and r["dp"] < 1e-7, res  # owned AND ghost entries agree with the serial run
        assert abs(r["diff"] - gdiffs[-1]) < 1e-8 * max(1.0, abs(gdiffs[-1])) and abs(r["vol"] - G._vol) < 1e-12
        assert r["its"] == res[0]["its"]  # the lock-step's decisions are all-reduced: the same on every rank
    gi = G.iteration_counts()
    for k in gi:
        assert abs(max(res[0]["its"][k]) - max(int(i) for i in gi[k])) <= 2, (k, res[0]["its"], gi)
    # every rank went through the same number of collective points
    assert len(set(world.allreduces)) == 1 and world.allreduces[0] > 0 and min(world.exchanges) > 0


def test_a_failed_pressure_solve_raises_on_every_rank(hip):
    """``ksp_error_if_not_converged`` on a partitioned job.  The convergence test's operands are all-reduced, so
    KSP_DIVERGED_ITS is reached at the same iteration on every rank and EVERY rank raises ``KSPConvergenceError`` -- no
    rank is left waiting in a collective for one that has already unwound (8 rank threads; a rank stuck in an exchange
    would run into the bounded waits of the harness and fail the test with a broken barrier instead)."""
    import oasisx_amd as ox
    from oasisx_amd import mesh as M
    from oasisx_amd.ksp import KSPConvergenceError
    from oracle import ipcs_oracle as O
    from tests.helpers import KRYLOV, on_boundary3, run_rank_threads

    nu, dt = 0.01, 0.005

    def rank_job(comm):
        mesh = M.create_box(comm, [[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]], [6, 6, 6])
        clock = {"t": 0.0}
        fns = [O.tg_u, O.tg_v, O.tg_w]
        bcs = [[ox.DirichletBC(lambda x, f=f: f(x, clock["t"], nu), ox.LocatorMethod.GEOMETRICAL, on_boundary3)] for f in fns]
        so = {k: dict(v) for k, v in KRYLOV.items()}
        so["pressure"].update(ksp_max_it=3, ksp_rtol=1e-14)  # (no pressure conditions: the solver sets error_if_not_converged)
        S = ox.FractionalStep_AB_CN(mesh, ("Lagrange", 2), ("Lagrange", 1), bcs_u=bcs, bcs_p=[], solver_options=so,
                                    options={"sell_window": 128})
        for i, f in enumerate(fns):
            S._u2[i].interpolate(lambda x, f=f: f(x, -dt, nu))
            S._u1[i].interpolate(lambda x, f=f: f(x, 0.0, nu))
        S._p.interpolate(lambda x: O.tg_p(x, -dt / 2, nu))
        clock["t"] = dt
        try:
            S.solve(dt, nu, max_iter=1)
        except KSPConvergenceError as e:
            torch.cuda.synchronize()
            return {"reasons": list(e.reasons), "iterations": list(e.iterations)}
        return None

    res, world = run_rank_threads(8, rank_job)
    assert all(r is not None for r in res), res
    assert all(r == res[0] for r in res) and res[0]["reasons"][0] == -3 and res[0]["iterations"][0] == 3, res
    assert len(set(world.allreduces)) == 1  # every rank left the solve at the same collective point
