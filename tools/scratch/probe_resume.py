import sys, numpy as np, torch
sys.path.insert(0, ".")
from tests.test_gpu_ksp_options import _ill_scaled, _matrix, _true_rel_residual
from oasisx_amd.fem import FieldStorage
from oasisx_amd.ksp import KSPSolver
for decades in (2.0, 3.5, 5.0):
    V, Acsr = _ill_scaled(2, 24, 2, decades, seed=3)
    A = _matrix(V, Acsr, symmetric=False)
    n = V.num_dofs
    rng = np.random.default_rng(5)
    for trial in range(3):
        b = Acsr @ rng.standard_normal(n)
        B = FieldStorage(n, 1, "cuda"); B.dev()[:, 0] = torch.from_numpy(b).cuda()
        for rtol in (1e-10, 1e-12, 1e-13, 1e-14, 3e-15):
            for merged in (False, True):
                ksp = KSPSolver(None, {"ksp_type": "bcgs", "pc_type": "jacobi", "ksp_rtol": rtol, "ksp_atol": 1e-300, "ksp_max_it": 2000, "ksp_bcgs_merged_reduction": merged})
                ksp.setOperators(A)
                X = FieldStorage(n, 1, "cuda")
                r = ksp.solve_block(B, X)[0]; res = ksp.last_result
                print(decades, trial, rtol, merged, "reason", r, "its", res.its[0], "rn/bn %.2e" % (res.rnorm[0]/res.bnorm[0]), "resumed", res.resumed[0], "true %.2e" % _true_rel_residual(Acsr, b, X.dev()[:,0].cpu().numpy()))
