"""TEST INFRASTRUCTURE ONLY -- Python driver of the C/OpenMP restatement (ipcs_cpu.c).

``CpuIPCS`` runs the IPCS step on the host in CSR, one velocity component at a time, exactly
in the order reference fracstep.py:660-696 drives DOLFINx/PETSc.  It is (a) validated against
``ipcs_oracle.py`` in tests/, (b) the timed ``cpu_baseline`` ("kind": "port") of bench.py.
"""
from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np

from . import ipcs_oracle as O

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def load():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libipcs_cpu.so")
        if not os.path.exists(path):  # normally built by __graft_entry__.build(); gcc is on every box
            import subprocess

            subprocess.run(["make", "-s", "-C", _HERE], check=False)
        if not os.path.exists(path):
            raise RuntimeError(f"{path} not built: run `make -C oracle`")
        _LIB = C.CDLL(path)
        _LIB.cpu_num_threads.restype = C.c_int
        for name in ("cpu_cg", "cpu_bicgstab"):
            getattr(_LIB, name).restype = C.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def reference_tensors(d, u_deg, p_deg):
    """Reference-element tensors the C kernels contract with (collapsed Gauss-Jacobi, exact)."""
    bary, w = O.simplex_quadrature(d, 4)
    phi, dphi = O.tabulate(d, u_deg, bary)
    psi, dpsi = O.tabulate(d, p_deg, bary)
    Tc = np.einsum("q,qk,qja,qi->kaji", w, phi, dphi, phi)
    Tp = np.einsum("q,qc,qia->ica", w, psi, dphi)
    Tg = np.einsum("q,qca,qi->ica", w, dpsi, phi)
    Td = np.einsum("q,qka,qi->ika", w, dphi, psi)
    return [np.ascontiguousarray(t) for t in (Tc, Tp, Tg, Td)]


class CpuIPCS:
    """Host-side IPCS step on CSR matrices.

    Args (all numpy, dof numbering arbitrary but consistent):
        geom: (nc, (d+1)*d+1) barycentric gradients (all d+1 rows) then |detJ|
        vd, qd: cell->dof tables (int32) of Vi and Q
        M, K: scipy CSR on Vi (same sorted pattern), Ap: scipy CSR on Q
        b0: (d, n_u) body-force vector, wq: int psi_i, vol: int 1
        bc_dofs: int32 dofs of the velocity Dirichlet condition (same for all components)
    """

    def __init__(self, d, u_deg, p_deg, geom, vd, qd, M, K, Ap, b0, wq, vol, bc_dofs, ksp):
        self.lib = load()
        self.d, self.u_deg, self.p_deg = d, u_deg, p_deg
        self.geom = np.ascontiguousarray(geom, dtype=np.float64)
        self.vd = np.ascontiguousarray(vd, dtype=np.int32)
        self.qd = np.ascontiguousarray(qd, dtype=np.int32)
        self.nc = self.vd.shape[0]
        self.nu_, self.nq = M.shape[0], Ap.shape[0]
        self.rp = np.ascontiguousarray(M.indptr, dtype=np.int64)
        self.ci = np.ascontiguousarray(M.indices, dtype=np.int32)
        self.Mv = np.ascontiguousarray(M.data, dtype=np.float64)
        self.Kv = np.ascontiguousarray(K.data, dtype=np.float64)
        self.Av = np.zeros_like(self.Mv)
        self.prp = np.ascontiguousarray(Ap.indptr, dtype=np.int64)
        self.pci = np.ascontiguousarray(Ap.indices, dtype=np.int32)
        self.pv = np.ascontiguousarray(Ap.data, dtype=np.float64)
        self.b0 = np.ascontiguousarray(b0, dtype=np.float64)
        self.wq, self.vol = np.ascontiguousarray(wq), float(vol)
        self.bc_dofs = np.ascontiguousarray(bc_dofs, dtype=np.int32)
        self.ksp = ksp  # {"rtol","atol","max_it","guess"}
        self.Tc, self.Tp, self.Tg, self.Td = reference_tensors(d, u_deg, p_deg)
        n, nq = self.nu_, self.nq
        z = lambda *s: np.zeros(s)  # noqa: E731
        self.u, self.u1, self.u2, self.uab = z(d, n), z(d, n), z(d, n), z(d, n)
        self.rhs1, self.b_first, self.tmp = z(d, n), z(d, n), z(d, n)
        self.p, self.ps, self.dp, self.b2 = z(nq), z(nq), z(nq), z(nq)
        self.work = z(6 * max(n, nq))
        self.dinvA, self.dinvM, self.dinvP = z(n), z(n), z(nq)
        self.lib.cpu_diag_inv(C.c_int64(n), _p(self.rp), _p(self.ci), _p(self.Mv), _p(self.dinvM))
        self.lib.cpu_diag_inv(C.c_int64(nq), _p(self.prp), _p(self.pci), _p(self.pv), _p(self.dinvP))
        self.its = {}
        self.threads = int(self.lib.cpu_num_threads())

    # -- helpers -----------------------------------------------------------------------
    def _solve(self, kind, n, rp, ci, v, dinv, b, x):
        its, rn = C.c_int(0), C.c_double(0.0)
        fn = self.lib.cpu_cg if kind == "cg" else self.lib.cpu_bicgstab
        k = self.ksp
        reason = fn(C.c_int64(n), _p(rp), _p(ci), _p(v), _p(dinv), _p(b), _p(x), C.c_double(k["rtol"]),
                    C.c_double(k["atol"]), C.c_int(k["max_it"]), C.c_int(int(k["guess"])), _p(self.work),
                    C.byref(its), C.byref(rn))
        return int(reason), int(its.value)

    def _spmv(self, v, x, y):
        self.lib.cpu_spmv(C.c_int64(self.nu_), _p(self.rp), _p(self.ci), _p(v), _p(x), _p(y))

    # -- the step (reference fracstep.py:660-696) ----------------------------------------
    def step(self, dt, nu, g):
        """``g``: (d, n_bc) Dirichlet values at ``bc_dofs`` for this time level."""
        lib, d, n, nq = self.lib, self.d, self.nu_, self.nq
        nnz = C.c_int64(self.Mv.shape[0])
        self.ps[:] = self.p
        # assemble_first (:411-472)
        lib.cpu_axpby(C.c_int64(d * n), C.c_double(1.5), _p(self.u1), C.c_double(-0.5), _p(self.u2), _p(self.uab))
        self.Av[:] = 0.0
        nd = self.vd.shape[1]
        lib.cpu_assemble_convection(C.c_int(d), C.c_int(nd), C.c_int64(self.nc), _p(self.geom), _p(self.vd),
                                    _p(self.Tc), C.c_int64(n), _p(self.uab), _p(self.rp), _p(self.ci), _p(self.Av))
        lib.cpu_matrix_phase(C.c_int(0), nnz, _p(self.Av), _p(self.Mv), _p(self.Kv), C.c_double(dt), C.c_double(nu))
        for i in range(d):
            self._spmv(self.Av, self.u1[i], self.b_first[i])
            self.b_first[i] += self.b0[i]
        lib.cpu_matrix_phase(C.c_int(1), nnz, _p(self.Av), _p(self.Mv), _p(self.Kv), C.c_double(dt), C.c_double(nu))
        lib.cpu_zero_rows(_p(self.rp), _p(self.ci), _p(self.Av), _p(self.bc_dofs), C.c_int64(self.bc_dofs.shape[0]))
        lib.cpu_diag_inv(C.c_int64(n), _p(self.rp), _p(self.ci), _p(self.Av), _p(self.dinvA))
        # velocity_tentative_assemble (:474-506)
        self.tmp[:] = 0.0
        ndq = self.qd.shape[1]
        lib.cpu_assemble_grad_vector(C.c_int(d), C.c_int(nd), C.c_int(ndq), C.c_int64(self.nc), _p(self.geom),
                                     _p(self.vd), _p(self.qd), _p(self.Tp), _p(self.ps), C.c_int64(n), _p(self.tmp))
        self.rhs1[:] = self.b_first + self.tmp
        # velocity_tentative_solve (:508-525)
        diff, its_t = 0.0, []
        for i in range(d):
            self.rhs1[i, self.bc_dofs] = g[i]
            old = self.u[i].copy()
            reason, its = self._solve("bcgs", n, self.rp, self.ci, self.Av, self.dinvA, self.rhs1[i], self.u[i])
            assert reason > 0, reason
            its_t.append(its)
            diff += float(np.linalg.norm(old - self.u[i]))
        # pressure_assemble (:527-551) + pressure_solve (:553-605)
        self.b2[:] = 0.0
        lib.cpu_assemble_div_vector(C.c_int(d), C.c_int(ndq), C.c_int(nd), C.c_int64(self.nc), _p(self.geom),
                                    _p(self.qd), _p(self.vd), _p(self.Td), C.c_int64(n), _p(self.u), _p(self.b2))
        self.b2 *= -1.0 / dt
        self.b2 -= self.b2.mean()
        reason, its_p = self._solve("cg", nq, self.prp, self.pci, self.pv, self.dinvP, self.b2, self.dp)
        assert reason > 0, reason
        self.dp -= float(self.wq @ self.dp) / self.vol
        self.ps[:] = self.p + self.dp
        # velocity_update (:607-658)
        self.tmp[:] = 0.0
        lib.cpu_assemble_grad_vector(C.c_int(d), C.c_int(nd), C.c_int(ndq), C.c_int64(self.nc), _p(self.geom),
                                     _p(self.vd), _p(self.qd), _p(self.Tg), _p(self.dp), C.c_int64(n), _p(self.tmp))
        its_c = []
        b3 = self.rhs1  # reuse
        for i in range(d):
            self._spmv(self.Mv, self.u[i], b3[i])
            b3[i] -= dt * self.tmp[i]
            reason, its = self._solve("cg", n, self.rp, self.ci, self.Mv, self.dinvM, b3[i], self.u[i])
            assert reason > 0, reason
            its_c.append(its)
        self.u2[:] = self.u1
        self.u1[:] = self.u
        self.p[:] = self.ps
        self.its = {"tentative": its_t, "pressure": [its_p], "update": its_c}
        return diff


def pack_geometry(G, adet):
    return np.ascontiguousarray(np.concatenate([G.reshape(G.shape[0], -1), adet[:, None]], axis=1))


def from_oracle(S: O.OracleFractionalStep, ksp):
    """CpuIPCS twin of a numpy-oracle solver (tests)."""
    F = S.F
    bc = S.bcs_u[0][0].dofs.astype(np.int32)
    cpu = CpuIPCS(F.d, F.u_deg, F.p_deg, pack_geometry(F.G, F.adet), F.vd, F.qd, S.M, S.K, S.Ap,
                  S.b0.T.copy(), S.wq, S.vol, bc, ksp)
    cpu.u[:], cpu.u1[:], cpu.u2[:] = S.u.T, S.u1.T, S.u2.T
    cpu.p[:] = S.p
    return cpu


def sell_to_csr_host(pattern, vals_list):
    """(indptr, indices, [values...]) host numpy arrays from a device SELL pattern (torch ops)."""
    import torch

    dev = pattern.device
    n = pattern.n_rows
    rl = pattern.row_len.to(torch.int64)
    rp = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    rp[1:] = torch.cumsum(rl, 0)
    r = torch.repeat_interleave(torch.arange(n, device=dev), rl)
    k = torch.arange(r.shape[0], device=dev) - rp[r]
    off = pattern.slice_ptr[r // 64] + (k // 2) * 128 + (r % 64) * 2 + (k % 2)
    del r, k
    cols = pattern.cols[off].cpu().numpy()
    vals = [v[off].cpu().numpy() for v in vals_list]
    return rp.cpu().numpy(), cols, vals


def run_cpu_baseline(S, clock, dt, nu, ksp, bc_values, gpu_step=None):
    """Time ONE step of the same workload, from the GPU solver's current state, on the host
    cores with the C/OpenMP port; optionally let the GPU take the same step and report the
    relative difference of the two results (a full-size parity data point)."""
    import scipy.sparse as sp
    import torch

    Vi, Q = S._Vi[0][0], S._Q
    mesh = S._mesh
    d = mesh.gdim
    t0 = time.perf_counter()
    rp, ci, (Mv, Kv) = sell_to_csr_host(Vi.pattern, [S._M.vals, S._K.vals])
    prp, pci, (pv,) = sell_to_csr_host(Q.pattern, [S._Ap.vals])
    n, nq = Vi.num_dofs, Q.num_dofs
    M = sp.csr_matrix((Mv, ci, rp), shape=(n, n))
    K = sp.csr_matrix((Kv, ci, rp), shape=(n, n))
    Ap = sp.csr_matrix((pv, pci, prp), shape=(nq, nq))
    geom = S._geom.cpu().numpy()  # rows lambda_1..d then |detJ|
    nc = geom.shape[0]
    Gd = geom[:, : d * d].reshape(nc, d, d)
    G = np.concatenate([-Gd.sum(axis=1, keepdims=True), Gd], axis=1)
    cpu = CpuIPCS(d, Vi.degree, Q.degree, pack_geometry(G, geom[:, d * d]), Vi.cell_dofs.cpu().numpy(),
                  Q.cell_dofs.cpu().numpy(), M, K, Ap, S._B0.dev()[:n].cpu().numpy().T.copy(),
                  S._wQ.cpu().numpy(), S._vol, S._bcs_u[0][0]._dofs, ksp)
    cpu.u[:] = S._U.dev()[:n].cpu().numpy().T
    cpu.u1[:] = S._U1.dev()[:n].cpu().numpy().T
    cpu.u2[:] = S._U2.dev()[:n].cpu().numpy().T
    cpu.p[:] = S._P.dev()[:nq, 0].cpu().numpy()
    cpu.dp[:] = S._DP.dev()[:nq, 0].cpu().numpy()
    t_setup = time.perf_counter() - t0
    clock["t"] += dt
    g = bc_values(clock["t"])
    t0 = time.perf_counter()
    cpu.step(dt, nu, g)
    t_step = time.perf_counter() - t0
    out = {"value": 1.0 / t_step, "unit": "steps/s", "cores": cpu.threads, "kind": "port",
           "sample": f"1 time step of the same workload (same mesh, state, Krylov settings) on the host: "
                     f"oracle/ipcs_cpu.c, OpenMP x{cpu.threads}, CSR, per-component solves as the reference",
           "seconds": t_step, "setup_seconds": t_setup, "krylov_iterations": cpu.its}
    if gpu_step is not None:
        clock["t"] -= dt
        gpu_step()
        ug = S._U1.dev()[:n].cpu().numpy().T
        pg = S._P.dev()[:nq, 0].cpu().numpy()
        out["gpu_vs_cpu_rel_l2_u"] = float(np.linalg.norm(ug - cpu.u1) / np.linalg.norm(cpu.u1))
        out["gpu_vs_cpu_rel_l2_p"] = float(np.linalg.norm(pg - cpu.p) / max(np.linalg.norm(cpu.p), 1e-300))
    return out
