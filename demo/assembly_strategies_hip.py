"""Assembly strategies on the HIP path -- the question of the reference's
demo/assembly_strategies.py (matrix-free "action" assembly vs cached matrices for the right-hand
sides of the fractional step), asked of the kernels behind ``FractionalStep_AB_CN``:

* the p*-gradient term of the tentative velocity (reference fracstep.py:487-506) and the
  divergence right-hand side of the pressure equation (:538-546), each as
  - "action": matrix-free cell kernels (``low_memory_version=True``: ox_assemble_grad_vector /
    ox_assemble_div_vector) and
  - "matvec": one pass over the pre-assembled rectangular operators
    (``low_memory_version=False``: ox_spmv_multi);
* the fused ``assemble_first`` (convection assembly + the six matrix passes + the d b_first
  mat-vecs of fracstep.py:432-469 in one kernel), for scale.

Usage: python demo/assembly_strategies_hip.py [--repeats 5] [--cells 30 25 23]
(the reference times a 30x25x23 unit cube, demo/assembly_strategies.py:221)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oasisx_amd as oasisx  # noqa: E402
from oasisx_amd import mesh as dmesh  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--repeats", type=int, default=5)
ap.add_argument("--cells", type=int, nargs=3, default=[30, 25, 23])
args = ap.parse_args()
dt, nu = 0.5, 0.3  # the arbitrary data of the reference demo (assembly_strategies.py:55-56)


def timed(fn, repeats):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))


rows = []
for P in (1, 2):
    mesh = dmesh.create_unit_cube(None, *args.cells)
    res = {}
    for strategy, lm in (("action", True), ("matvec", False)):
        S = oasisx.FractionalStep_AB_CN(mesh, ("Lagrange", P), ("Lagrange", 1), bcs_u=[[], [], []], bcs_p=[],
                                        solver_options=None, options={"low_memory_version": lm})
        for i in range(3):
            S._u1[i].interpolate(lambda x, i=i: np.sin(x[i]) + x[(i + 1) % 3] ** 2)
            S._u2[i].interpolate(lambda x, i=i: np.cos(x[i]))
            S._u[i].interpolate(lambda x, i=i: x[i] * x[(i + 2) % 3])
        S._ps.interpolate(lambda x: x[0] * x[1] + x[2])
        res["first"] = timed(lambda: S.assemble_first(dt, nu), args.repeats)
        res[strategy, "gradp"] = timed(S.velocity_tentative_assemble, args.repeats)
        res[strategy, "div"] = timed(lambda: S.pressure_assemble(dt), args.repeats)
        ndofs = S._n_u
        del S
    rows.append((P, ndofs, res))

print(f"mesh {args.cells[0]}x{args.cells[1]}x{args.cells[2]} x 6 tetrahedra, median of {args.repeats} (ms)")
print(f"{'P':>2} {'dofs/comp':>10} | {'grad p*: action':>16} {'matvec':>8} | {'div u: action':>14} {'matvec':>8} | "
      f"{'fused assemble_first':>20}")
for P, n, r in rows:
    print(f"{P:>2} {n:>10} | {r['action', 'gradp']:>16.3f} {r['matvec', 'gradp']:>8.3f} | "
          f"{r['action', 'div']:>14.3f} {r['matvec', 'div']:>8.3f} | {r['first']:>20.3f}")
