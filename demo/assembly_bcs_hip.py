#!/usr/bin/env python3
"""Left-hand side / right-hand side assembly of the tentative-velocity step with Dirichlet rows -- the
question of the reference's demo/assembly_bcs.py:131-203 (its "Oasis approach": A = M/dt - C/2 - nu K/2,
b = A u_1 by a mat-vec, then A <- -A + 2M/dt and identity rows; against assembling matrix and vector
separately), asked of the HIP path:

* fused      ``assemble_first``: convection assembly, the six matrix passes, the d mat-vecs b = A u_1 and
             the final A in ONE kernel (csrc/ox_assemble.hip), then ``zero_rows`` + ``set_bc``;
* separate   the same matrix from the same kernel, but the right-hand side by mat-vecs with the stored
             operators, as the reference does: b = -(A u_1) + (2/dt) M u_1 (two passes over 6 GB
             matrices at 128^3 instead of none).

Both give the same vector (checked); the times are HIP-event medians.

    python demo/assembly_bcs_hip.py [-N 40] [--degree 2] [--repeats 5]
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run_assembly_bcs(N: int = 40, degree: int = 2, repeats: int = 5, dt: float = 0.1, nu: float = 0.1) -> dict:
    import torch

    import oasisx_amd as ox
    from oasisx_amd import _lib
    from oasisx_amd import mesh as M

    mesh = M.create_unit_cube(None, N, N, N)

    def walls(x):  # the reference's inlet + walls: every face but x = 1
        on = np.isclose(x[0], 0.0)
        for k in (1, 2):
            on |= np.isclose(x[k], 0.0) | np.isclose(x[k], 1.0)
        return on

    bcs = [[ox.DirichletBC((lambda x, c=c: np.sin(np.pi * x[1]) * (c == 0)), ox.LocatorMethod.GEOMETRICAL, walls)]
           for c in range(3)]
    S = ox.FractionalStep_AB_CN(mesh, ("Lagrange", degree), ("Lagrange", 1), bcs_u=bcs, bcs_p=[],
                                options={"low_memory_version": True})
    for i in range(3):
        S._u1[i].interpolate(lambda x, i=i: np.sin(x[i]) + x[(i + 1) % 3] ** 2)
        S._u2[i].interpolate(lambda x, i=i: np.cos(x[i]) * x[(i + 2) % 3])
    lib, st = S._lib, _lib.current_stream()
    n = S._n_u * 3

    def timed(fn):
        fn()
        ts = []
        for _ in range(repeats):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return float(np.median(ts))

    def bcs_on_rhs():
        for i in range(3):
            for bc in S._bcs_u[i]:
                bc.apply(S._b_first[i].x)

    t_fused = timed(lambda: S.assemble_first(dt, nu))  # matrix + b_first + identity rows
    t_bc = timed(bcs_on_rhs)
    b_fused = S._BFIRST.dev().clone()
    # the reference's order: the mat-vec comes BEFORE the identity rows -- reproduce b from the stored
    # operators: A_final = -Ar + (2/dt) M  =>  Ar u_1 = -(A_final u_1) + (2/dt) M u_1 on the un-BC'd rows
    tmp, tmp2 = S._WRK, S._B3

    def rhs_separate():
        S._A.mult(S._U1.dev(), tmp.dev(), 3)
        S._M.mult(S._U1.dev(), tmp2.dev(), 3)
        _lib.check(lib.ox_axpby(n, -1.0, tmp.ptr(), 2.0 / dt, tmp2.ptr(), tmp.ptr(), st), "ox_axpby")

    t_sep = timed(rhs_separate)
    free = torch.ones(S._n_u, dtype=torch.bool, device="cuda")
    free[S._bcs_u[0][0]._rows_dev.to(torch.int64)] = False  # rows of A that zeroRowsLocal turned into identity
    diff = float((tmp.dev()[free] - b_fused[free]).abs().max() / b_fused[free].abs().max())
    return {"N": N, "degree": degree, "dofs_per_component": S._n_u, "nnz": S._A.pattern.nnz,
            "fused_lhs_rhs_ms": t_fused, "set_bc_ms": t_bc, "separate_rhs_matvecs_ms": t_sep,
            "separate_total_ms": t_fused + t_sep, "rhs_rel_diff": diff}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("-N", type=int, default=40)  # the reference's 40^3 cube (demo/assembly_bcs.py:376-380)
    ap.add_argument("--degree", type=int, nargs="+", default=[1, 2])
    ap.add_argument("--repeats", type=int, default=5)
    a = ap.parse_args()
    print(f"unit cube {a.N}^3 x 6 tetrahedra, median of {a.repeats} (ms)")
    print(f"{'P':>2} {'dofs/comp':>10} {'nnz':>11} | {'fused LHS+RHS':>14} {'set_bc':>8} | {'+ RHS by mat-vecs':>18} | rel. diff")
    for deg in a.degree:
        r = run_assembly_bcs(a.N, deg, a.repeats)
        print(f"{deg:>2} {r['dofs_per_component']:>10} {r['nnz']:>11} | {r['fused_lhs_rhs_ms']:>14.3f} {r['set_bc_ms']:>8.3f} | "
              f"{r['separate_rhs_matvecs_ms']:>18.3f} | {r['rhs_rel_diff']:.1e}")
