#!/usr/bin/env python3
"""Convergence study of the IPCS solver on the 2-D Taylor-Green vortex, HIP path.

The study the reference runs in CI (its demo/taylor_green.py with ``-N 8 -N 16 -N 32 -dt 0.005``,
.github/workflows/tests.yml:59): exact Dirichlet velocity on the whole boundary of [-1, 1]^2, no
pressure condition, space-time L2 errors of u and p per mesh and the observed rates.  This file is a
harness of our own around ``oasisx_amd`` -- a function per concern, importable by the tests
(``run_taylor_green``) -- not a transcription of the reference script.

    python demo/taylor_green_hip.py 8 16 32 --dt 0.005 --T 1.0 [--krylov] [--rotational] [--out DIR]
"""
from __future__ import annotations

import argparse
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

DIRECT = {k: {"ksp_type": "preonly", "pc_type": "lu"} for k in ("tentative", "pressure", "scalar")}
KRYLOV = {"tentative": {"ksp_type": "bcgs", "pc_type": "jacobi", "ksp_rtol": 1e-10, "ksp_atol": 1e-30},
          "pressure": {"ksp_type": "cg", "pc_type": "jacobi", "ksp_rtol": 1e-10, "ksp_atol": 1e-30},
          "scalar": {"ksp_type": "cg", "pc_type": "jacobi", "ksp_rtol": 1e-10, "ksp_atol": 1e-30}}


class TaylorGreen2D:
    """The analytic fields u = (-cos(pi x) sin(pi y), sin(pi x) cos(pi y)) e^{-2 nu pi^2 t},
    p = -(cos(2 pi x) + cos(2 pi y)) / 4 e^{-4 nu pi^2 t}; ``now`` is the time the callables read."""

    def __init__(self, nu: float):
        self.nu = nu
        self.now = 0.0

    def decay(self, power: int, t: float | None = None) -> float:
        return math.exp(-power * self.nu * math.pi ** 2 * (self.now if t is None else t))

    def velocity(self, comp: int, t: float | None = None):
        sign, a, b = ((-1.0, 0, 1), (1.0, 1, 0))[comp]
        return lambda x: sign * np.cos(np.pi * x[a]) * np.sin(np.pi * x[b]) * self.decay(2, t)

    def pressure(self, t: float | None = None):
        return lambda x: -0.25 * (np.cos(2 * np.pi * x[0]) + np.cos(2 * np.pi * x[1])) * self.decay(4, t)


def build_solver(N: int, field: TaylorGreen2D, degree_u: int, solver_options, low_memory: bool, rotational: bool,
                 degree_p: int = 1):
    import oasisx_amd as ox
    from oasisx_amd import mesh as M

    mesh = M.create_rectangle(None, [[-1.0, -1.0], [1.0, 1.0]], [N, N])
    # tag every exterior facet and hand the tags to the Dirichlet conditions (the topological route)
    fdim = mesh.topology.dim - 1
    boundary = np.sort(M.exterior_facet_indices(mesh.topology))
    tags = M.meshtags(mesh, fdim, boundary, np.full(boundary.shape, 1, dtype=np.int32))
    bcs = [[ox.DirichletBC(field.velocity(c), ox.LocatorMethod.TOPOLOGICAL, (tags, 1))] for c in range(2)]
    assert degree_u > degree_p  # reference demo/taylor_green.py:111
    solver = ox.FractionalStep_AB_CN(mesh, ("Lagrange", degree_u), ("Lagrange", degree_p), bcs_u=bcs, bcs_p=[],
                                     rotational=rotational, solver_options=solver_options,
                                     options={"low_memory_version": low_memory})
    return mesh, solver


def run_taylor_green(N: int, dt: float = 0.005, T: float = 1.0, nu: float = 0.01, degree_u: int = 2,
                     solver_options=None, low_memory: bool = False, rotational: bool = False,
                     out_dir: str | None = None, degree_p: int = 1) -> dict:
    """March from t = 0 to T on the N x N mesh; returns h, the space-time L2 errors
    sqrt(dt sum_n ||e^n||^2) of u and p and the Krylov iteration counts of the last step."""
    from oasisx_amd import fem

    field = TaylorGreen2D(nu)
    mesh, solver = build_solver(N, field, degree_u, solver_options or DIRECT, low_memory, rotational, degree_p)
    for c in range(2):  # two velocity levels and the staggered pressure (t = -dt, 0, -dt/2)
        solver._u2[c].interpolate(field.velocity(c, -dt))
        solver._u1[c].interpolate(field.velocity(c, 0.0))
    solver._p.interpolate(field.pressure(-dt / 2.0))
    writers = []
    if out_dir:
        from oasisx_amd import io

        writers = [io.VTXWriter(mesh.comm, os.path.join(out_dir, f"u_{N}.bp"), [solver.u], engine="BP4"),
                   io.VTXWriter(mesh.comm, os.path.join(out_dir, f"p_{N}.bp"), [solver._p], engine="BP4")]
    steps = int(round(T / dt))
    sq_u = sq_p = 0.0
    for n in range(1, steps + 1):
        field.now = n * dt
        solver.solve(dt, nu, max_iter=1)
        sq_u += sum(fem.assemble_l2_error_sq(solver._u[c], field.velocity(c)) for c in range(2))
        sq_p += fem.assemble_l2_error_sq(solver._p, field.pressure(field.now - dt / 2.0))
        for w in writers:
            w.write(field.now)
    for w in writers:
        w.close()
    cells = np.arange(mesh.topology.index_map(mesh.topology.dim).size_local)
    return {"N": N, "h": float(mesh.h(mesh.topology.dim, cells).max()), "steps": steps,
            "error_u": math.sqrt(dt * sq_u), "error_p": math.sqrt(dt * sq_p),
            "iterations": solver.iteration_counts()}


def observed_rates(results, key):
    r = sorted(results, key=lambda e: -e["h"])
    return [math.log(a[key] / b[key]) / math.log(a["h"] / b["h"]) for a, b in zip(r, r[1:])]


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("N", type=int, nargs="+", help="cells per direction, one run each")
    ap.add_argument("--dt", type=float, default=0.005)
    ap.add_argument("--T", type=float, default=1.0)
    ap.add_argument("--nu", type=float, default=0.01)
    ap.add_argument("-u", "--degree-u", type=int, default=2, help="velocity degree (reference demo: -u)")
    ap.add_argument("-p", "--degree-p", type=int, default=1, help="pressure degree (reference demo: -p); built pairs: 2-1, 3-2")
    ap.add_argument("--krylov", action="store_true", help="Jacobi-BiCGStab / Jacobi-CG at 1e-10 instead of 'preonly lu'")
    ap.add_argument("--low-memory", action="store_true")
    ap.add_argument("--rotational", action="store_true")
    ap.add_argument("--out", default=None, help="directory for a VTK time series of u and p")
    a = ap.parse_args(argv)
    results = []
    for N in a.N:
        r = run_taylor_green(N, a.dt, a.T, a.nu, a.degree_u, KRYLOV if a.krylov else DIRECT, a.low_memory,
                             a.rotational, a.out, a.degree_p)
        results.append(r)
        print(f"N = {N:4d}  h = {r['h']:.5f}  ||e_u|| = {r['error_u']:.4e}  ||e_p|| = {r['error_p']:.4e}  "
              f"iterations {r['iterations']}")
    if len(results) > 1:
        print("rates u:", " ".join(f"{v:.2f}" for v in observed_rates(results, "error_u")))
        print("rates p:", " ".join(f"{v:.2f}" for v in observed_rates(results, "error_p")))
    return results


if __name__ == "__main__":
    main()
