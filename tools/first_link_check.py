#!/usr/bin/env python3
"""FIRST CONTACT with a link: what to run, once, on a node with 2 (or 4, 8) GPUs before trusting a scaling figure.

No round of this build had a multi-GPU node: the partitioned path has run between ranks on ONE device only (process
rehearsals over IPC windows and the host-staged transport, 8 rank threads in one process), RCCL only as one-rank
self-loops.  This script walks every device transport through the checks ADVICE r05 asks for, on real ranks:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
        tools/first_link_check.py                       # one rank per GPU, torch.distributed over RCCL
    ... tools/first_link_check.py --backend gloo        # rehearsal of the script itself: the ranks share GPU 0

For each transport -- ``rccl`` (the default of a job with an RCCL communicator), ``p2p`` with the CONSERVATIVE release
protocol (the library's default), ``p2p`` with the fast one (``OX_P2P_RELEASE=fast``), ``host`` (callback transport) --:
  1. the halo self-test of both spaces at set-up (``check_halo``: every ghost dof receives its owner's coordinates);
  2. a bit-exact all-reduce: 15 values of very different magnitudes per rank; every rank must hold the same bits, and for
     the window transport they must equal the host's sum in rank order;
  3. two whole time steps (3-D Taylor-Green, P2-P1) against the serial run: owned and ghost entries to 1e-8 / 1e-7,
     equal iteration counts on all ranks;
  4. 200 back-to-back exchanges + all-reduces under the bounded waits (``ox_dist_status`` must stay clean), timed.
Rank 0 prints one JSON object per transport and a final verdict line; a failed check raises on the rank that saw it.
Only after this has passed with ``p2p fast`` on real links should ``ox_dist_set_p2p_release(plan, 0)`` / OX_TRANSPORT=auto
become anyone's default.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def key(c):
    import numpy as np

    return [tuple(r) for r in np.round((np.asarray(c) + 1.0) * float(1 << 35)).astype(np.int64).tolist()]


def run(N, comm, steps):
    import oasisx_amd as ox
    from oasisx_amd import mesh as M
    from oracle import ipcs_oracle as O  # (test infrastructure: the analytic fields only)
    from tests.helpers import KRYLOV, on_boundary3

    nu, dt = 0.01, 0.005
    mesh = M.create_box(comm, [[-1.0] * 3, [1.0] * 3], [N, N, N])
    clock = {"t": 0.0}
    fns = [O.tg_u, O.tg_v, O.tg_w]
    bcs = [[ox.DirichletBC(lambda x, f=f: f(x, clock["t"], nu), ox.LocatorMethod.GEOMETRICAL, on_boundary3)] for f in fns]
    opts = {k: dict(v, ksp_initial_guess_nonzero=True) for k, v in KRYLOV.items()}
    S = ox.FractionalStep_AB_CN(mesh, ("Lagrange", 2), ("Lagrange", 1), bcs_u=bcs, bcs_p=[], solver_options=opts,
                                options={"sell_window": 128})
    for i, f in enumerate(fns):
        S._u2[i].interpolate(lambda x, f=f: f(x, -dt, nu))
        S._u1[i].interpolate(lambda x, f=f: f(x, 0.0, nu))
    S._p.interpolate(lambda x: O.tg_p(x, -dt / 2, nu))
    for _ in range(steps):
        clock["t"] += dt
        S.solve(dt, nu, max_iter=1)
    return S


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("-N", type=int, default=10)
    ap.add_argument("--transports", default=None, help="comma list of rccl,p2p,p2p-fast,host (default: all the backend allows)")
    args = ap.parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    dev = local % ndev if args.backend == "nccl" else 0  # (gloo rehearsal: every rank on GPU 0)
    if args.backend == "nccl" and world > ndev:
        raise SystemExit(f"first_link_check: {world} ranks over RCCL need {world} GPUs (found {ndev}); use --backend gloo to rehearse")
    torch.cuda.set_device(dev)
    if args.backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo")
    from oasisx_amd import _lib
    from oasisx_amd.parallel import init_comm

    lib = _lib.load()
    todo = args.transports.split(",") if args.transports else (["rccl"] if args.backend == "nccl" else []) + ["p2p", "p2p-fast", "host"]
    G = run(args.N, None, 2)  # the serial run, on every rank
    torch.cuda.synchronize()
    lu = {k: i for i, k in enumerate(key(G._Vi[0][0].x.cpu().numpy()))}
    lq = {k: i for i, k in enumerate(key(G._Q.x.cpu().numpy()))}
    ug, pg = G._U1.dev().cpu().numpy(), G._P.dev().cpu().numpy()[:, 0]
    verdict = {}
    for name in todo:
        os.environ["OX_TRANSPORT"] = name.split("-")[0]
        os.environ["OX_P2P_TIMEOUT_S"] = "60"
        if name == "p2p-fast":
            os.environ["OX_P2P_RELEASE"] = "fast"
        else:
            os.environ.pop("OX_P2P_RELEASE", None)
        comm = init_comm()
        S = run(args.N, comm, 2)  # (1) check_halo of both spaces runs inside attach_comm at set-up
        torch.cuda.synchronize()
        Vi, Q = S._Vi[0][0], S._Q
        active = dict(comm.active)
        # (3) the fields against the serial run, owned and ghost entries
        iu = np.asarray([lu[k] for k in key(Vi.x.cpu().numpy())])
        iq = np.asarray([lq[k] for k in key(Q.x.cpu().numpy())])
        du = float(np.abs(S._U1.dev().cpu().numpy() - ug[iu]).max())
        dp = float(np.abs(S._P.dev().cpu().numpy()[:, 0] - pg[iq]).max())
        assert du < 1e-8 and dp < 1e-7, (name, rank, du, dp)
        its = {k: [int(i) for i in v] for k, v in S.iteration_counts().items()}
        # (2) bit-exact all-reduce on the velocity plan
        rng = np.random.default_rng(11)
        vals = rng.standard_normal((world, 15)) * 10.0 ** rng.integers(-8, 8, size=(world, 15))
        buf = torch.from_numpy(vals[rank].copy()).cuda()
        _lib.check(lib.ox_allreduce_sum(Vi.dist, _lib.ptr(buf), 15, _lib.current_stream()), "ox_allreduce_sum")
        torch.cuda.synchronize()
        got = buf.cpu().numpy()
        gathered = [None] * world
        dist.all_gather_object(gathered, (got.tobytes(), its))
        assert all(g[0] == gathered[0][0] for g in gathered), f"{name}: the ranks hold different all-reduce bits"
        assert all(g[1] == gathered[0][1] for g in gathered), f"{name}: iteration counts differ between ranks"
        want = np.zeros(15)
        for r in range(world):
            want = want + vals[r]
        rank_order = bool(np.array_equal(got, want))
        if name.startswith("p2p"):
            assert rank_order, f"{name}: the window all-reduce is not the sum in rank order"
        assert np.abs(got - want).max() <= 1e-12 * np.abs(vals).max()
        # (4) exchanges back to back under the bounded waits, timed (collective)
        times = comm.time_transports(Vi, reps=200)
        S._Vi[0][0].check_halo()
        S._Q.check_halo()
        for V in (Vi, Q):
            _lib.check(lib.ox_dist_status(V.dist), "ox_dist_status")
        res = {"transport": name, "active": active, "max_abs_du": du, "max_abs_dp": dp, "iterations": its,
               "allreduce_bits_equal_on_all_ranks": True, "allreduce_is_the_rank_order_sum": rank_order,
               "exchange_us": times.get(name.split("-")[0]) or times, "ranks": world,
               "devices": sorted({int(x) for x in _all(dist, dev, world)})}
        verdict[name] = "passed"
        if rank == 0:
            print(json.dumps(res), flush=True)
        del S
        torch.cuda.empty_cache()
        dist.barrier()
    if rank == 0:
        print(json.dumps({"first_link_check": verdict, "backend": args.backend,
                          "note": "all device transports that ran passed the halo self-test, the bit-exact all-reduce and the "
                                  "two-step comparison with the serial run" + ("" if args.backend == "nccl" else
                                  " -- REHEARSAL on one GPU (gloo): no link was crossed")}), flush=True)
    dist.destroy_process_group()


def _all(dist, dev, world):
    out = [None] * world
    dist.all_gather_object(out, dev)
    return out


if __name__ == "__main__":
    main()
