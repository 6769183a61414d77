"""``FractionalStep_AB_CN`` -- the IPCS fractional-step solver (Adams-Bashforth convection,
Crank-Nicolson diffusion) with the Python surface of reference src/oasisx/fracstep.py:29-705,
running on hand-written HIP kernels (liboasisx_hip.so) instead of DOLFINx assembly + PETSc.

Data layout on the device
  * velocity-like fields are (n_u, gdim) interleaved blocks, so the gdim component systems,
    which share one matrix (reference fracstep.py:274,521,634), are multiplied and solved
    together: one pass over the matrix per SpMV instead of gdim;
  * M, K, A share one SELL-64 pattern (reference fracstep.py:293-294), Ap has its own;
  * ``low_memory_version=True`` evaluates the p*-, div(u)- and grad(phi)-terms matrix-free
    (fracstep.py:485-497,537-538,612-634); ``False`` pre-assembles the 3*gdim rectangular
    operators (fracstep.py:311-315,332-336,348-352,392-404) as two SELL patterns with gdim values
    per entry and applies them with one SpMV pass each (fracstep.py:499-502,540-542,642).

Deviations from the reference, by necessity of the platform:
  * ``preonly``+``lu`` is mapped to a tightly converged Krylov solve (ksp.py);
  * without pressure BCs the reference overrides the pressure solver with MUMPS
    (fracstep.py:562-576); here the configured solver (CG + Jacobi) runs on the mean-free
    right-hand side -- the mass-weighted shift that follows makes the result unique.
"""
from __future__ import annotations

import ctypes as C
import logging

import numpy as np
import torch

from . import _lib
from .fem import FieldStorage, Function, FunctionSpace, VectorFunctionSpace, cell_geometry
from .ksp import KSPSolver
from .la import MultiSellMatrix, SellMatrix

__all__ = ["FractionalStep_AB_CN"]


def _degree(element):
    if isinstance(element, FunctionSpace):
        return element.degree
    fam = str(element[0]).lower()
    if fam not in ("lagrange", "p", "cg"):
        raise ValueError(f"unsupported element family {element[0]!r}")
    return int(element[1])


def _phase(fn):
    """roctx range around a phase method (``ox_range_push/pop``: rocprofv3 --marker-trace segments a trace by
    phase; two no-op C calls otherwise)."""
    import functools

    name = ("oasisx::" + fn.__name__).encode()

    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        self._lib.ox_range_push(name)
        try:
            return fn(self, *a, **k)
        finally:
            self._lib.ox_range_pop()

    return wrapped


class _RotationalUpdate:
    """The linear form int (p + dp - xi nu div(u)) q dx of the rotational pressure update
    (reference fracstep.py:240): M_Q (p + dp) - xi nu int div(u) q."""

    def __init__(self, solver):
        self.s = solver

    def assemble_rhs_into(self, B):
        s = self.s
        lib, st = s._lib, _lib.current_stream()
        Vi, Q = s._Vi[0][0], s._Q
        nq = s._n_q
        tmp = s._B2  # free at this point of the step: the pressure system has been solved
        _lib.check(lib.ox_axpby(nq, 1.0, s._P.ptr(), 1.0, s._DP.ptr(), tmp.ptr(), st), "ox_axpby")
        s._projector_p._A.mult(tmp.dev(), B.dev(), 1)
        _lib.check(lib.ox_assemble_div_vector(Q.degree, Vi.degree, C.byref(s._cells), _lib.ptr(Vi.cell_dofs),
                                              C.byref(s._adj_q), Q.n_owned, s._U.ptr(), 1.0, tmp.ptr(), st),
                   "ox_assemble_div_vector")
        _lib.check(lib.ox_axpby(Q.n_owned, 1.0, B.ptr(), -s._xi * s._nu, tmp.ptr(), B.ptr(), st), "ox_axpby")


class FractionalStep_AB_CN:
    """
    Create the fractional step solver with Adam-Bashforth linearization
    of the convective term, and Crank-Nicholson time discretization.

    Args:
        mesh: The computational domain
        u_element: ``("Lagrange", k)`` for each velocity component (k = 1, 2)
        p_element: ``("Lagrange", 1)`` for the pressure
        bcs_u: list of Dirichlet BCs for each component of the velocity
        bcs_p: list of pressure BCs
        rotational: If True, use rotational form of pressure update
        solver_options: dict with keys ``'tentative'``, ``'pressure'``, ``'scalar'`` leading
            to PETSc-style option dicts (see :mod:`oasisx_amd.ksp`)
        jit_options: accepted for compatibility, ignored (nothing is JIT compiled)
        options: ``"low_memory_version"`` accepted (see module docstring)
        body_force: per direction a constant, a callable ``f(x)``, a Function of the component space or a
            :class:`oasisx_amd.function.Expression` of fields (assembled once, as in the reference)
    """

    def __init__(self, mesh, u_element, p_element, bcs_u, bcs_p, rotational: bool = False,
                 solver_options: dict | None = None, jit_options: dict | None = None,
                 body_force=None, options: dict | None = None):
        self._lib = _lib.load()  # fails loudly when the HIP library is missing
        self._mesh = mesh
        gdim = mesh.geometry.dim
        self._gdim = gdim
        dev = mesh.device
        u_deg, p_deg = _degree(u_element), _degree(p_element)
        # Taylor-Hood pairs P2-P1 (every BASELINE configuration) and P3-P2 on triangles and tetrahedra (the reference demo's
        # -u 3 -p 2: demo/taylor_green.py:82-83,111; one GPU); equal order P1-P1 (BASELINE configs[1])
        if (u_deg, p_deg) not in ((1, 1), (2, 1), (3, 2)):
            raise NotImplementedError(f"Lagrange P{u_deg}-P{p_deg}: the built pairs are P1-P1, P2-P1 and P3-P2")
        window = int((options or {}).get("sell_window", 4096))

        # ---- spaces (reference fracstep.py:186-216) ----------------------------------------
        comm = getattr(mesh, "comm", None)
        self._comm = comm
        part = None
        if comm is not None and getattr(comm, "size", 1) > 1:
            from .parallel import MeshPartition

            part = MeshPartition(mesh, comm.rank, comm.size, comm, faces=(u_deg == 3))  # (P3: face / cell dofs have owners)
        self._part = part
        # options["spmv_windows"]: the LDS-window stream of the velocity pattern (k_spmv_win).  Default: ON for meshes
        # that are not lattices (Z-order numbering as it is: the mat-vecs of a refined Delaunay mesh run 1.3-1.9 x
        # faster), OFF on lattice meshes, where it needs the brick order of the numbering and was measured a wash on the
        # whole step at 128^3 (profiles/r04_spmv_window_experiments.txt).  One GPU only.
        from .fem import mesh_is_lattice

        lattice = mesh_is_lattice(mesh)
        # (mesh-partitioned operators too, in the Z-order numbering: their window blocks are listed interior-first and
        # the overlapped mat-vec multiplies the two halves around the halo exchange; the brick order is one-GPU only)
        windows = bool((options or {}).get("spmv_windows", not lattice)) and dev.type == "cuda" \
            and (not lattice or (part is None and u_deg >= 2))
        self._spmv_windows = windows
        self._lattice = lattice
        # options["assemble_row_blocks"]: the row kernels -- assemble_first, M, K, Ap -- run as ONE launch over the slices in
        # storage order (ox_assemble_first_blocks) instead of one launch per width bin; bit-identical either way.
        # Default: on for meshes that are not lattices (refined Delaunay mesh at the bench size: nine bins, 95.9 GB of
        # HBM traffic and 18.3 ms per assemble_first -> one launch, 31.9 GB, 13.5 ms); off on lattice meshes, whose
        # three bins re-fetch little (24.6 GB) and whose LDS-filling blocks pay the dispatch gap between row blocks
        # (128^3: 7.44 ms with the bins, 7.83 ms in one launch; tools/af_bench.py, profiles/r05_assemble_row_blocks.txt)
        self._row_blocks = bool((options or {}).get("assemble_row_blocks", not lattice))
        Vi = u_element if isinstance(u_element, FunctionSpace) else FunctionSpace(mesh, u_deg, window=window, part=part,
                                                                                  brick=windows and lattice)
        if isinstance(p_element, FunctionSpace):
            Q = p_element
        else:
            Q = Vi if u_deg == p_deg else FunctionSpace(mesh, p_deg, window=window, part=part)
        if part is not None:
            Vi.attach_comm(comm)
            if Q is not Vi:
                Q.attach_comm(comm)
        self._V = VectorFunctionSpace(Vi, gdim)
        self._Vi = [self._V.sub(i).collapse() for i in range(gdim)]
        self._Q = Q
        nu_, nq_ = Vi.n_local, Q.n_local  # owned + ghost rows of every field block
        self._n_u, self._n_q = nu_, nq_
        self._no_u, self._no_q = Vi.n_owned, Q.n_owned
        self._du, self._dq = Vi.dist, Q.dist  # halo plans (None on one GPU)

        def block():
            return FieldStorage(nu_, gdim, dev)

        self._U, self._U1, self._U2, self._UAB = block(), block(), block(), block()
        self._RHS1, self._B0, self._BFIRST, self._WRK = block(), block(), block(), block()
        self._B3 = block()
        comp = range(gdim)
        self._sol_u = Function(self._V, "u", storage=self._U, comp=None)
        self._u = [Function(Vi, f"u{i}", self._U, i) for i in comp]
        self._u1 = [Function(Vi, f"u_{i}1", self._U1, i) for i in comp]
        self._u2 = [Function(Vi, f"u_{i}2", self._U2, i) for i in comp]
        self._uab = [Function(Vi, f"u_{i}ab", self._UAB, i) for i in comp]
        self._rhs1 = [Function(Vi, f"rhs1_{i}", self._RHS1, i) for i in comp]
        self._b0 = [Function(Vi, f"b0_{i}", self._B0, i) for i in comp]
        self._b_first = [Function(Vi, f"b_first_{i}", self._BFIRST, i) for i in comp]
        self._wrk_vel = [Function(Vi, f"wrk_{i}", self._WRK, i) for i in comp]
        self._wrk_comp = self._wrk_vel[0]  # reference fracstep.py:204 (work vector of one component)
        self._b3 = Function(Vi, "b3", self._B3, 0)  # reference fracstep.py:341

        self._PS, self._P, self._DP, self._B2 = (FieldStorage(nq_, 1, dev) for _ in range(4))
        self._ps = Function(Q, "ps", self._PS, 0)
        self._p = Function(Q, "p", self._P, 0)
        self._dp = Function(Q, "dp", self._DP, 0)
        self._b2 = Function(Q, "b2", self._B2, 0)

        # ---- boundary conditions (reference fracstep.py:196-200,218-227) ---------------------
        from .fem import shared_marker_evaluations

        self._bcs_u = bcs_u
        self._bcs_p = bcs_p
        with shared_marker_evaluations():  # the gdim components usually share one marker object
            for bc_i in self._bcs_u:
                for bc in bc_i:
                    bc.create_bc(Vi)
            for bcp in self._bcs_p:  # reference fracstep.py:219-227
                bcp.create_bcs(Vi, Q)
        if len(self._bcs_p) > 0:
            self._p_surf = [[bcp.rhs(i) for bcp in self._bcs_p] for i in range(gdim)]

        # ---- solvers (reference fracstep.py:229-255) -------------------------------------
        solver_options = {} if solver_options is None else solver_options
        self._solver_u = KSPSolver(mesh.comm, solver_options.get("tentative"), prefix="tentative_velocity")
        self._solver_p = KSPSolver(mesh.comm, solver_options.get("pressure"), prefix="pressure_correction")
        self._solver_c = KSPSolver(mesh.comm, solver_options.get("scalar"), prefix="velocity_update")
        self._rotational = bool(rotational)
        self._projector_p = None
        self._xi = 0.5 if rotational else None  # reference fracstep.py:238
        self._nu = 1.0 if rotational else None

        if options is None:
            options = {}
        self._low_memory = options.get("low_memory_version", True)
        self._options = dict(options)
        if body_force is None:
            body_force = (0.0,) * gdim
        # per component: a float, a Constant, a spatial expression f(x) (x: (3, npts) -> (npts,)) or a Function on
        # the component space -- `force * v * dx` is assembled ONCE (reference fracstep.py:284-289,387-390)
        from .function import Expression

        self._body_force = [f if (callable(f) or isinstance(f, (Function, Expression))) else float(f) for f in body_force]
        if len(self._body_force) != gdim:
            raise ValueError(f"body_force: {gdim} components expected")

        self._compile_and_allocate_forms()
        self._preassemble()

        if self._rotational:  # Projector(p + dp - xi nu div(u), Q) (fracstep.py:237-247)
            from .function import Projector

            self._projector_p = Projector(_RotationalUpdate(self), Q, bcs=[],
                                          petsc_options=solver_options.get("scalar"), jit_options=jit_options)
        # reference fracstep.py:270-275
        self._solver_p.setOperators(self._Ap)
        self._solver_p.setOptions(self._Ap)
        self._solver_c.setOperators(self._M)
        self._solver_u.setOperators(self._A)
        self._solver_u.setOptions(self._A)
        self.timings = {}

    # ------------------------------------------------------------------------------------
    def _compile_and_allocate_forms(self):
        """Device mesh data and matrix allocation (reference fracstep.py:277-358: nothing is
        compiled here -- the element kernels are the templated HIP kernels of ox_assemble.hip)."""
        mesh = self._mesh
        Vi, Q = self._Vi[0][0], self._Q
        if getattr(Vi, "native", None) is not None:  # the library's own geometry (kernel cell order)
            self._geom = Vi.native.nmesh.geom
        else:
            self._geom = cell_geometry(mesh, Vi.local_cells)
        self._cells = _lib.ox_cells(mesh.gdim, 0, int(self._geom.shape[0]), self._geom.data_ptr())
        self._adj_u = Vi.adj.struct()
        self._adj_q = Q.adj.struct()
        if not self._low_memory:
            from .fem import build_rect_pattern

            self._pat_vq, self._pos_vq, self._pw_vq = build_rect_pattern(Vi, Q)
            self._pat_qv, self._pos_qv, self._pw_qv = build_rect_pattern(Q, Vi)
            self._p_vdxi_Mat = MultiSellMatrix(self._pat_vq, mesh.gdim, "p_vdxi")
            self._grad_p_Mat = MultiSellMatrix(self._pat_vq, mesh.gdim, "grad_p")
            self._divu_Mat = MultiSellMatrix(self._pat_qv, mesh.gdim, "divu")
        # LDS-window stream of the velocity pattern (M, K, A share it): single-GPU operators of a degree-2 space
        def windows(space, what, split=0):
            # (an optional storage level: a failed build -- out of device memory in its sorts -- leaves the space on the
            # lane = row kernels, with a warning; results are the same either way)
            try:
                space.build_windows(split)
            except _lib.OasisxHipError as e:
                logging.getLogger("oasisx").warning("LDS-window stream of the %s pattern not built (%s): its mat-vecs run "
                                                    "on the lane = row kernels", what, e)
        if self._spmv_windows:
            # (its mat-vecs run with gdim right-hand sides: blocks beyond the LDS budget of a three-column launch are cut
            # in two; the one-column pressure pattern keeps whole blocks -- csrc/ox_setup.hip build_windows)
            windows(Vi, "velocity", 2176 if mesh.gdim == 3 else 3072)
        # the pressure matrix too where it has no pair-slot stream to lose (meshes that are not lattices carry no
        # value dictionary): refined Delaunay mesh, 2.4 M P1 rows: 120 -> 86 us per mat-vec, 0.73 of the HBM peak
        if self._options.get("spmv_windows_pressure", self._spmv_windows and not self._lattice) and mesh.device.type == "cuda":
            windows(Q, "pressure")
        self._M = SellMatrix(Vi.pattern, symmetric=True, name="M")
        self._K = SellMatrix(Vi.pattern, symmetric=True, name="K")
        self._A = SellMatrix(Vi.pattern, symmetric=False, name="A")
        self._Ap = SellMatrix(Q.pattern, symmetric=True, name="Ap")

    def _assemble_matrix(self, kind, V: FunctionSpace, adj_struct, Mat: SellMatrix):
        if self._row_blocks and V.pattern.n_row_blocks > 0:  # one launch over the slices in storage order
            nblk, bptr, ent = V.pattern.blocks_args()
            _lib.check(self._lib.ox_assemble_matrix_blocks(kind, V.degree, C.byref(self._cells), _lib.ptr(V.cell_dofs),
                                                           C.byref(adj_struct), _lib.ptr(V.adj.adj_pos), V.adj.pw,
                                                           Mat.ref(), nblk, bptr, ent, _lib.current_stream()),
                       "ox_assemble_matrix_blocks")
        else:
            nb, bptr, bsl, bw = V.pattern.bins_args()
            _lib.check(self._lib.ox_assemble_matrix(kind, V.degree, C.byref(self._cells), _lib.ptr(V.cell_dofs),
                                                    C.byref(adj_struct), _lib.ptr(V.adj.adj_pos), V.adj.pw,
                                                    Mat.ref(), nb, bptr, bsl, bw, _lib.current_stream()),
                       "ox_assemble_matrix")
        Mat.version += 1

    def _preassemble(self):
        """Time-independent operators (reference fracstep.py:360-409)."""
        lib, st = self._lib, _lib.current_stream()
        Vi, Q = self._Vi[0][0], self._Q
        dev = self._mesh.device
        self._assemble_matrix(0, Vi, self._adj_u, self._M)  # mass        (:373)
        self._assemble_matrix(1, Vi, self._adj_u, self._K)  # stiffness   (:375)
        self._assemble_matrix(1, Q, self._adj_q, self._Ap)  # pressure Laplacian (:379)
        if len(self._bcs_p) > 0:  # assemble_matrix(..., bcs): BC rows and columns -> identity
            is_bc = torch.zeros(Q.n_local, dtype=torch.uint8, device=dev)
            for bcp in self._bcs_p:
                is_bc[bcp._dofs_dev.to(torch.int64)] = 1
            _lib.check(lib.ox_zero_rows_cols(self._Ap.ref(), _lib.ptr(is_bc), 1.0, st), "ox_zero_rows_cols")
            self._Ap.version += 1
        if self._options.get("value_dictionary", True):
            # M and Ap never change again: 1-byte value codes where <= 256 distinct values (la.freeze)
            self._M.freeze()
            self._K.freeze(pairs="never")  # read by the fused assemble_first only: never multiplied
            self._Ap.freeze()
        if not self._low_memory:  # the rectangular operators (:392-404)
            for fam, Mat, R_, C_, adj_, pos_, pw_ in ((0, self._p_vdxi_Mat, Vi, Q, self._adj_u, self._pos_vq, self._pw_vq),
                                                    (1, self._grad_p_Mat, Vi, Q, self._adj_u, self._pos_vq, self._pw_vq),
                                                    (2, self._divu_Mat, Q, Vi, self._adj_q, self._pos_qv, self._pw_qv)):
                _lib.check(lib.ox_assemble_rect(fam, R_.degree, C_.degree, C.byref(self._cells), C.byref(adj_),
                                                _lib.ptr(pos_), pw_, Mat.ref(), st), "ox_assemble_rect")
                if self._options.get("value_dictionary", True):
                    Mat.freeze()
        # int phi_r dx on both spaces: body force vector (:387-390), mean of phi (:585-590)
        self._wV = torch.zeros(Vi.n_owned, dtype=torch.float64, device=dev)
        self._wQ = torch.zeros(Q.n_owned, dtype=torch.float64, device=dev)
        _lib.check(lib.ox_assemble_weights(Vi.degree, C.byref(self._cells), C.byref(self._adj_u), Vi.n_owned,
                                           _lib.ptr(self._wV), st), "ox_assemble_weights")
        _lib.check(lib.ox_assemble_weights(Q.degree, C.byref(self._cells), C.byref(self._adj_q), Q.n_owned,
                                           _lib.ptr(self._wQ), st), "ox_assemble_weights")
        B0 = self._B0.dev()
        for i, f in enumerate(self._body_force):
            if isinstance(f, Function):  # a field of the component space: int f v dx = M f
                if getattr(f.function_space, "scalar", f.function_space) is not Vi:
                    raise ValueError("body_force: a Function must live on the velocity component space")
                src = f._storage.rdev()[:, 0 if f._comp is None else f._comp].contiguous().unsqueeze(1)
                tmp = torch.zeros(Vi.n_local, 1, dtype=torch.float64, device=dev)
                self._M.mult(src, tmp, 1)
                B0[: Vi.n_owned, i] = tmp[: Vi.n_owned, 0]
            elif not isinstance(f, float):
                # a spatial expression (or a pointwise expression of fields, function.Expression): tabulated at quadrature
                # points, summed by ox_assemble_load_vector
                from .function import load_vector, metadata_points

                q = metadata_points({"quadrature_degree": self._options.get("body_force_quadrature_degree")}, Vi.degree)
                B0[: Vi.n_owned, i] = load_vector(Vi, f, self._geom, q)[: Vi.n_owned]
            else:
                B0[: Vi.n_owned, i] = self._wV * float(f)
        vol = float(self._wQ.sum().item())  # assemble_scalar(1*dx) + allreduce (:581-584)
        self._vol = vol if self._part is None else float(self._comm.allreduce(vol))

    # ------------------------------------------------------------------------------------
    @_phase
    def assemble_first(self, dt: float, nu: float):
        """Reference fracstep.py:411-472: A = M/dt + C/2 + nu K/2 and the part of the RHS that
        depends on the previous time step, b_k = (M/dt - C/2 - nu K/2) u_k^{n-1} + b0_k."""
        lib, st = self._lib, _lib.current_stream()
        n = self._n_u * self._gdim
        # u_ab = 1.5 u_1 - 0.5 u_2 (:432-434)
        _lib.check(lib.ox_axpby(n, 1.5, self._U1.rptr(), -0.5, self._U2.rptr(), self._UAB.ptr(), st), "ox_axpby")
        Vi = self._Vi[0][0]
        # With a nonzero initial guess the tentative solve starts from u (= u1 bit for bit unless someone
        # wrote to it since the last step): its first mat-vec A @ u1 falls out of the fused kernel's
        # epilogue (same entry order and operations as the SpMV).  Kept in the block of b3, free until
        # velocity_update.
        self._AU1_valid = False
        want_au = bool(self._solver_u._options.get("ksp_initial_guess_nonzero", False)) and \
            str(self._solver_u._options.get("ksp_type", "")).lower() != "preonly"
        if self._row_blocks and Vi.pattern.n_row_blocks > 0:
            # ONE launch over the slices in storage order (round 5): the rows of a cell meet in one L2 instead of being
            # torn apart into the launches of up to ten width bins (options["assemble_row_blocks"]; bit-identical)
            nblk, bptr, ent = Vi.pattern.blocks_args()
            _lib.check(lib.ox_assemble_first_blocks(Vi.degree, C.byref(self._cells), _lib.ptr(Vi.cell_dofs),
                                                    C.byref(self._adj_u), _lib.ptr(Vi.adj.adj_pos), Vi.adj.pw,
                                                    self._A.ref(), self._M.ref(), self._K.ref(),
                                                    self._UAB.rptr(), self._U1.rptr(), self._B0.rptr(), self._BFIRST.ptr(),
                                                    float(dt), float(nu), nblk, bptr, ent, st,
                                                    self._B3.ptr() if want_au else None), "ox_assemble_first_blocks")
        else:
            nb, bptr, bsl, bw = Vi.pattern.bins_args()
            _lib.check(lib.ox_assemble_first_au(Vi.degree, C.byref(self._cells), _lib.ptr(Vi.cell_dofs),
                                                C.byref(self._adj_u), _lib.ptr(Vi.adj.adj_pos), Vi.adj.pw,
                                                self._A.ref(), self._M.ref(), self._K.ref(),
                                                self._UAB.rptr(), self._U1.rptr(), self._B0.rptr(), self._BFIRST.ptr(),
                                                float(dt), float(nu), nb, bptr, bsl, bw, st,
                                                self._B3.ptr() if want_au else None), "ox_assemble_first")
        self._A.version += 1
        # outlet terms int h n_i dv/dx_i ds (:445-446, :461-465)
        for bcp in self._bcs_p:
            bcp.update_bc()
            bcp.add_surface_terms(self._BFIRST)
        # NOTE (reference :470): rows of the FIRST component's BCs only
        for bcu in self._bcs_u[0]:  # identity rows; (A @ u1)[row] = u1[row] there, in the same launch
            self._A.zero_rows(bcu._rows_dev, 1.0, self._B3.ptr() if want_au else None,
                              self._U1.rptr() if want_au else None, self._gdim)
        self._AU1_valid = want_au

    @_phase
    def velocity_tentative_assemble(self):
        """rhs1_k = b_first_k + int p* dv/dx_k (reference fracstep.py:474-506)."""
        Vi, Q = self._Vi[0][0], self._Q
        if not self._low_memory:  # P_i.mult(ps) (:499-502)
            self._p_vdxi_Mat.mult(False, self._PS.ptr(), self._BFIRST.ptr(), 1.0, self._RHS1.ptr())
            return
        _lib.check(self._lib.ox_assemble_grad_vector(0, Vi.degree, Q.degree, C.byref(self._cells),
                                                     _lib.ptr(Q.cell_dofs), C.byref(self._adj_u), Vi.n_owned,
                                                     self._PS.ptr(), self._BFIRST.ptr(), 1.0, self._RHS1.ptr(),
                                                     _lib.current_stream()), "ox_assemble_grad_vector")

    @_phase
    def velocity_tentative_solve(self):
        """Apply the Dirichlet values to the RHS and solve the gdim tentative-velocity systems
        (reference fracstep.py:508-525).  Returns (diff, errors)."""
        lib, st = self._lib, _lib.current_stream()
        gdim = self._gdim
        for i in range(gdim):
            for bc in self._bcs_u[i]:
                bc.apply(self._rhs1[i].x)
        n = self._n_u * gdim
        _lib.check(lib.ox_axpby(n, 1.0, self._U.rptr(), 0.0, None, self._WRK.ptr(), st), "ox_axpby")
        ax0 = None
        if getattr(self, "_AU1_valid", False):  # one use per assemble_first, and only if u still is u1
            self._AU1_valid = False
            # u was copied into u1 at the end of the last step (solve()); since then neither block has been
            # written by one of this class's phases (they clear the token; this class reads the two blocks through
            # rptr() meanwhile) nor from outside: EVERY hand-out of a writable view or pointer -- host check-outs,
            # dev(), ptr(): DirichletBC.apply, KSPSolver.solve/solve_block, interpolate, S._U.dev()[...] = ... --
            # bumps FieldStorage.generation.  No device compare, no host synchronisation
            same = getattr(self, "_u_is_u1", None) == (self._U.generation, self._U1.generation)
            if self._part is not None and getattr(self._comm, "size", 1) > 1:
                # every rank must take the same branch: the skipped mat-vec carries a halo exchange
                same = self._comm.allreduce(0.0 if same else 1.0, op="max") == 0.0
            if same:
                ax0 = self._B3
            elif not getattr(self, "_shortcut_note", False):
                # (said once: a monitor that reads u through dev() / host() between steps makes every step pay one
                # mat-vec more than it has to)
                self._shortcut_note = True
                logging.getLogger("oasisx").info(
                    "tentative solve: u was handed out writable since the last step (FieldStorage.dev() / ptr() / host()): "
                    "the A u1 product of assemble_first is not reused; read-only access goes through rdev() / rptr() / rhost()")
        self._u_is_u1 = None  # the solve below writes u
        errors = np.asarray(self._solver_u.solve_block(self._RHS1, self._U, ax0=ax0), dtype=np.int32)
        # diff = sum_i || u_i^old - u_i ||_2 (:523-524)
        _lib.check(lib.ox_axpby(n, 1.0, self._WRK.ptr(), -1.0, self._U.ptr(), self._WRK.ptr(), st), "ox_axpby")
        out = (C.c_double * 4)()
        _lib.check(lib.ox_dot(self._no_u, gdim, self._WRK.ptr(), self._WRK.ptr(), out, self._du, st), "ox_dot")
        diff = float(sum(np.sqrt(out[i]) for i in range(gdim)))
        return diff, errors

    @_phase
    def pressure_assemble(self, dt: float):
        """b2 = -(1/dt) int div(u) q (reference fracstep.py:527-551)."""
        Vi, Q = self._Vi[0][0], self._Q
        if not self._low_memory:  # sum_i D_i.mult(u_i), scaled (:540-546)
            self._divu_Mat.mult(True, self._U.ptr(), None, -1.0 / float(dt), self._B2.ptr())
        else:
            _lib.check(self._lib.ox_assemble_div_vector(Q.degree, Vi.degree, C.byref(self._cells),
                                                        _lib.ptr(Vi.cell_dofs), C.byref(self._adj_q), Q.n_owned,
                                                        self._U.ptr(), -1.0 / float(dt), self._B2.ptr(),
                                                        _lib.current_stream()), "ox_assemble_div_vector")
        for bcp in self._bcs_p:  # homogeneous Dirichlet condition on the correction (:549-550)
            bcp.apply_homogeneous(self._b2.x)

    @_phase
    def pressure_solve(self, nu: float | None = None, rotational: bool = False):
        """Solve the pressure-correction problem (reference fracstep.py:553-605)."""
        lib, st = self._lib, _lib.current_stream()
        logger = logging.getLogger("oasisx")
        nq, nqo = self._n_q, self._no_q
        if len(self._bcs_p) == 0:
            # The reference switches its pressure solver to MUMPS LU here and sets ksp_error_if_not_converged (:565-572).
            # The direct solver has no device counterpart (the configured Krylov method runs on the
            # mean-free right-hand side); the error behaviour is kept: a failed solve raises instead of returning.
            if "ksp_error_if_not_converged" not in self._solver_p._options:  # (an explicit setting of the caller stands)
                self._solver_p.updateOptions({"ksp_error_if_not_converged": 1})
            # nullspace.remove(b2): subtract the arithmetic mean (:573-574)
            _lib.check(lib.ox_remove_mean(nqo, nqo, self._B2.ptr(), None, float(self._Q.num_dofs_global),
                                          self._dq, st), "ox_remove_mean")
        converged = self._solver_p.solve_block(self._B2, self._DP)[0]
        if len(self._bcs_p) == 0:
            logger.debug("Making sure that mean of phi is 0 with lack of pressure conditions")
            # dp -= (int dp dx) / (int 1 dx) (:579-591)
            _lib.check(lib.ox_remove_mean(nqo, nq, self._DP.ptr(), _lib.ptr(self._wQ), self._vol, self._dq, st),
                       "ox_remove_mean")
        if self._projector_p is not None:  # rotational update (:593-602)
            if nu is None:
                raise RuntimeWarning("Kinematic viscosity not set for rotational pressure correction")
            self._nu = float(nu)
            error = self._projector_p.solve(assemble_rhs=True)
            assert int(error) > 0
            _lib.check(lib.ox_axpby(nq, 1.0, self._projector_p._X.ptr(), 0.0, None, self._PS.ptr(), st), "ox_axpby")
        else:
            # ps = p + dp (:604)
            _lib.check(lib.ox_axpby(nq, 1.0, self._P.ptr(), 1.0, self._DP.ptr(), self._PS.ptr(), st), "ox_axpby")
        return converged

    @_phase
    def velocity_update(self, dt) -> np.ndarray:
        """M u_k = M u*_k - dt int dphi/dx_k v (reference fracstep.py:607-658; un-BC'd mass
        matrix, no Dirichlet re-imposition, exactly as the reference)."""
        lib, st = self._lib, _lib.current_stream()
        Vi, Q = self._Vi[0][0], self._Q
        gdim = self._gdim
        self._AU1_valid = False  # b3 overwrites the block that held A u1 (assemble_first)
        self._u_is_u1 = None
        # M u* is kept (in the work block of the tentative solve, free here): with a nonzero initial
        # guess it is the solver's first mat-vec A x0, which is then skipped
        MU = self._WRK
        self._M.mult(self._U.dev(), MU.dev(), gdim)
        if not self._low_memory:  # b3 = M u* - dt * G_i.mult(dp) (:642-645)
            self._grad_p_Mat.mult(False, self._DP.ptr(), MU.ptr(), -float(dt), self._B3.ptr())
        else:
            _lib.check(lib.ox_assemble_grad_vector(1, Vi.degree, Q.degree, C.byref(self._cells),
                                                   _lib.ptr(Q.cell_dofs), C.byref(self._adj_u), Vi.n_owned,
                                                   self._DP.ptr(), MU.ptr(), -float(dt), self._B3.ptr(), st),
                       "ox_assemble_grad_vector")
        return np.asarray(self._solver_c.solve_block(self._B3, self._U, ax0=MU), dtype=np.int32)

    def solve(self, dt: float, nu: float, max_error: float = 1e-12, max_iter: int = 10):
        """Propagate the splitting scheme one time step (reference fracstep.py:660-696)."""
        lib, st = self._lib, _lib.current_stream()
        inner_it = 0
        diff = 1e8
        nq, n = self._n_q, self._n_u * self._gdim
        _lib.check(lib.ox_axpby(nq, 1.0, self._P.ptr(), 0.0, None, self._PS.ptr(), st), "ox_axpby")
        for bcu in self._bcs_u:
            for bc in bcu:
                bc.update_bc()
        self.assemble_first(dt, nu)
        while inner_it < max_iter and diff > max_error:
            inner_it += 1
            self.velocity_tentative_assemble()
            diff, errors = self.velocity_tentative_solve()
            assert (errors > 0).all(), f"tentative velocity solve failed: {errors}"
            self.pressure_assemble(dt)
            error_p = self.pressure_solve(nu=nu)
            assert int(error_p) > 0, f"pressure solve failed: {error_p}"
        errors_c = self.velocity_update(dt)
        self._last_errors = (errors, error_p, errors_c)
        # u2 <- u1, u1 <- u, p <- ps (:689-693)
        _lib.check(lib.ox_axpby(n, 1.0, self._U1.ptr(), 0.0, None, self._U2.ptr(), st), "ox_axpby")
        _lib.check(lib.ox_axpby(n, 1.0, self._U.ptr(), 0.0, None, self._U1.ptr(), st), "ox_axpby")
        _lib.check(lib.ox_axpby(nq, 1.0, self._PS.ptr(), 0.0, None, self._P.ptr(), st), "ox_axpby")
        self._u_is_u1 = (self._U.generation, self._U1.generation)  # u1 is a bit copy of u from here on
        for d_ in (self._du, self._dq):  # partitioned runs: a peer wait that timed out anywhere in the step fails it
            if d_ is not None:
                _lib.check(lib.ox_dist_status(d_), "ox_dist_status (halo exchange / all-reduce of this step)")
        return diff

    @property
    def u(self):
        """The velocity as a blocked vector function (reference fracstep.py:698-705); it shares
        storage with the component functions ``_u[i]``."""
        return self._sol_u

    def iteration_counts(self):
        return {"tentative": self._solver_u.iterations, "pressure": self._solver_p.iterations,
                "update": self._solver_c.iterations}
