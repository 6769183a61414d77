#!/usr/bin/env python3
"""Taylor-Green convergence demo on the HIP path -- the counterpart of reference
demo/taylor_green.py (same arguments, same set-up, same error table).

    python demo/taylor_green_hip.py -N 8 -N 16 -N 32 -dt 0.005

Differences from the reference script, all forced by the platform: the pressure initial condition
and the exact fields are Python callables instead of UFL expressions, no BP4 output is written,
and ``preonly``+``lu`` maps to tightly converged Krylov solves (oasisx_amd/ksp.py).
"""
import argparse
import logging
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oasisx_amd as oasisx  # noqa: E402
from oasisx_amd import fem  # noqa: E402
from oasisx_amd import mesh as dmesh  # noqa: E402


class U:
    def __init__(self, t, nu):
        self.t = t
        self.nu = nu

    def eval_x(self, x):
        return -np.cos(np.pi * x[0]) * np.sin(np.pi * x[1]) * np.exp(-2.0 * self.nu * np.pi ** 2 * float(self.t))

    def eval_y(self, x):
        return np.cos(np.pi * x[1]) * np.sin(np.pi * x[0]) * np.exp(-2.0 * self.nu * np.pi ** 2 * float(self.t))


parser = argparse.ArgumentParser(description="Taylor-Green convergence demo",
                                 formatter_class=argparse.ArgumentDefaultsHelpFormatter)
parser.add_argument("-N", "--refinement", type=int, dest="Ns", action="append", required=True,
                    help="The number of elements in x and y direction")
parser.add_argument("-T0", "--T-start", dest="T_start", type=float, default=0, help="Start time of simulation")
parser.add_argument("-T1", "--T-end", dest="T_end", type=float, default=1, help="End time of simulation")
parser.add_argument("-dt", dest="dt", type=float, default=0.1, help="Time step")
parser.add_argument("-nu", dest="nu", type=float, default=0.01, help="Kinematic viscosity")
parser.add_argument("-u", dest="u_deg", type=int, default=2, help="Degree of velocity space")
parser.add_argument("-p", dest="p_deg", type=int, default=1, help="Degree of pressure space")
parser.add_argument("-lm", "--low-memory", dest="lm", action="store_true", default=False)
parser.add_argument("-r", "--rotational", dest="rot", action="store_true", default=False)
parser.add_argument("-o", "--output-dir", dest="outdir", default=None,
                    help="write u and p every step (VTK series, the stand-in for the reference's u.bp / p.bp)")
inputs = parser.parse_args()
logger = logging.getLogger("Oasisx")
logger.setLevel(logging.INFO)

dt, nu = inputs.dt, inputs.nu
assert inputs.T_start < inputs.T_end
T_end, T_start = inputs.T_end, inputs.T_start
num_steps = int((T_end - T_start) // dt)
assert inputs.u_deg > inputs.p_deg
el_u, el_p = ("Lagrange", inputs.u_deg), ("Lagrange", inputs.p_deg)
options = {"low_memory_version": inputs.lm}
solver_options = {k: {"ksp_type": "preonly", "pc_type": "lu"} for k in ("tentative", "pressure", "scalar")}

space_errors = np.zeros((2, len(inputs.Ns)))
hs = np.zeros(len(inputs.Ns))
for n, N in enumerate(inputs.Ns):
    mesh = dmesh.create_rectangle(None, [[-1, -1], [1, 1]], [N, N], cell_type=dmesh.CellType.triangle)
    dim = mesh.topology.dim - 1
    mesh.topology.create_connectivity(dim, dim + 1)
    facets = dmesh.exterior_facet_indices(mesh.topology)
    value = np.int32(3)
    values = np.full_like(facets, value, dtype=np.int32)
    sort = np.argsort(facets)
    facet_tags = dmesh.meshtags(mesh, dim, facets[sort], values[sort])

    u_time = fem.Constant(mesh, T_start)
    p_time = fem.Constant(mesh, T_start - dt / 2.0)
    u_ex = U(t=u_time, nu=nu)
    bcx = oasisx.DirichletBC(u_ex.eval_x, oasisx.LocatorMethod.TOPOLOGICAL, (facet_tags, value))
    bcy = oasisx.DirichletBC(u_ex.eval_y, oasisx.LocatorMethod.TOPOLOGICAL, (facet_tags, value))
    solver = oasisx.FractionalStep_AB_CN(mesh, el_u, el_p, bcs_u=[[bcx], [bcy]], bcs_p=[],
                                         rotational=inputs.rot, solver_options=solver_options,
                                         options=options, body_force=None)
    u_time.value = T_start - dt
    solver._u2[0].interpolate(u_ex.eval_x)
    solver._u2[1].interpolate(u_ex.eval_y)
    u_time.value = T_start
    solver._u1[0].interpolate(u_ex.eval_x)
    solver._u1[1].interpolate(u_ex.eval_y)

    def man_p(x):
        return -0.25 * (np.cos(2 * np.pi * x[0]) + np.cos(2 * np.pi * x[1])) * np.exp(
            -4 * np.pi ** 2 * nu * float(p_time))

    solver._p.interpolate(man_p)
    vtxu = vtxp = None
    if inputs.outdir:  # reference: VTXWriter(mesh.comm, "u.bp", [solver.u], engine="BP4")
        from oasisx_amd import io

        vtxu = io.VTXWriter(mesh.comm, os.path.join(inputs.outdir, f"u_{N}.bp"), [solver.u], engine="BP4")
        vtxp = io.VTXWriter(mesh.comm, os.path.join(inputs.outdir, f"p_{N}.bp"), [solver._p], engine="BP4")
    error_space_time = np.zeros((2, num_steps))
    u_time.value = T_start
    for i in range(num_steps):
        u_time.value += dt
        p_time.value += dt
        solver.solve(dt, nu, max_iter=1)
        error_u = (fem.assemble_l2_error_sq(solver._u[0], u_ex.eval_x)
                   + fem.assemble_l2_error_sq(solver._u[1], u_ex.eval_y))
        error_p = fem.assemble_l2_error_sq(solver._p, man_p)
        if vtxu is not None:
            vtxp.write(float(p_time.value))
            vtxu.write(float(u_time.value))
        error_space_time[:, i] = [error_u, error_p]
    if vtxu is not None:
        vtxu.close()
        vtxp.close()
    hmax = float(np.max(mesh.h(mesh.topology.dim, np.arange(mesh.topology.index_map(mesh.topology.dim).size_local))))
    space_time_u_L2 = np.sqrt(dt * np.sum(error_space_time[0, :]))
    space_time_p_L2 = np.sqrt(dt * np.sum(error_space_time[1, :]))
    logger.info(f"{hmax=} {space_time_u_L2=} {space_time_p_L2=}")
    hs[n] = hmax
    space_errors[:, n] = [space_time_u_L2, space_time_p_L2]

order = np.argsort(hs)[::-1]
hs[:] = hs[order]
space_errors[0, :] = space_errors[0, order]
space_errors[1, :] = space_errors[1, order]
if len(hs) > 1:
    rate_u = np.log(space_errors[0, 1:] / space_errors[0, :-1]) / np.log(hs[1:] / hs[:-1])
    rate_p = np.log(space_errors[1, 1:] / space_errors[1, :-1]) / np.log(hs[1:] / hs[:-1])
    logger.info(f"Convergence rates u: {rate_u}")
    logger.info(f"Convergence rates p: {rate_p}")
