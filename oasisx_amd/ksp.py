"""``KSPSolver``: the reference's thin PETSc-KSP wrapper (reference src/oasisx/ksp.py:14-91),
backed by the device-resident Jacobi-CG / Jacobi-BiCGStab of ``liboasisx_hip.so``.

Option mapping (PETSc string keys, as the reference passes them):
  ksp_type  cg -> CG;  bcgs -> BiCGStab;  gmres/other/unset -> CG if the operator is flagged
            symmetric else BiCGStab;  preonly (+ pc_type lu/cholesky) -> the same Krylov
            method run to ``direct_rtol`` (there is no sparse LU on the device; the converged
            reason reported is KSP_CONVERGED_ITS = 4 as PETSc's preonly does)
  pc_type   jacobi; none (PETSc's PCNONE: the identity -- dinv = 1 through the same kernels).  Anything else (ilu,
            hypre, gamg, ...; lu/cholesky outside preonly) runs as jacobi AND SAYS SO: every option this path cannot
            honour -- another preconditioner, an unknown Krylov type, a key it does not know -- is reported once per
            solver with a warning on the ``oasisx`` logger (the reference forwards any key to PETSc, ksp.py:38-53)
  ksp_cg_single_reduction  PETSc's option name: the Chronopoulos-Gear recurrences with one merged
            reduction (one all-reduce) per iteration; default: true on mesh-partitioned operators,
            false (PETSc's default, faster there) on a single GPU
  ksp_cg_merged_reduction  (extension) one-column "cg" solves: OX_KSP_CG_MERGED -- alpha and beta from ONE synchronisation
            point (one all-reduce), three kernels per iteration instead of five, no extra vectors; same Krylov space,
            same test on the true |D^-1 r|.  Default: true on mesh-partitioned operators (one all-reduce per iteration
            without the two extra vectors of the single-reduction form) and on one GPU up to 2^20 rows (launch-bound
            iterations: small meshes), false above
  ksp_bcgs_merged_reduction  (extension; also selected by ksp_type ibcgs / pipebcgs / fbcgsr, PETSc's reduced-
            synchronisation BiCGStab variants): two merged reductions (all-reduces) per iteration instead of
            three -- omega, rho and the residual norm from one reduction behind the second mat-vec
            (OX_KSP_BCGS_MERGED); default: true on mesh-partitioned operators, false on a single GPU
  ksp_rtol, ksp_atol, ksp_max_it, ksp_initial_guess_nonzero: as in PETSc (defaults 1e-5,
            1e-50, 10000, false -> the solution vector is zeroed before the solve)
  ksp_divtol  as in PETSc (default 1e4): a column whose |D^-1 r| reaches divtol |D^-1 b| ends with
            KSP_DIVERGED_DTOL (-4) instead of running to ksp_max_it
  ksp_error_if_not_converged  as in PETSc (the reference sets it for its pressure solver, fracstep.py:570):
            a negative converged reason raises ``KSPConvergenceError`` (PETSc: error 91) instead of being returned
  ksp_cg_fold_blocks, ksp_run_ahead  (extensions: schedule knobs of THIS solver, passed to the library per call)
            blocks of the folded one-column CG update kernels (0: five kernels per iteration; default: one per
            compute unit, or the environment's OX_CG_FOLD_BLOCKS read once by this module) and whether one-column
            solves queue one batch ahead of the state the host reads (default true; OX_KSP_RUN_AHEAD)
  ksp_bcgs_restarts (extension): BiCGStab restarts allowed after a rho/omega breakdown; default 0
            for an explicit "bcgs" (PETSc's KSPBCGS stops with DIVERGED_BREAKDOWN), 5 when the
            method stands in for a direct solver, which cannot break down (e.g. a cold start with
            identity-row Dirichlet conditions makes rho = 0 after one iteration)
"""
from __future__ import annotations

import ctypes as C
import logging
import os
import typing

import torch

from . import _lib
from .fem import FieldStorage, Function
from .la import SellMatrix

__all__ = ["KSPSolver", "KSPConvergenceError"]


class KSPConvergenceError(RuntimeError):
    """``ksp_error_if_not_converged``: the solve ended with a negative KSPConvergedReason (PETSc raises error 91,
    "KSPSolve has not converged", at the same place: reference ksp.py:76 with fracstep.py:570)."""

    def __init__(self, prefix, reasons, its):
        self.reasons, self.iterations = list(reasons), list(its)
        names = {-3: "DIVERGED_ITS", -4: "DIVERGED_DTOL", -5: "DIVERGED_BREAKDOWN", -9: "DIVERGED_NANORINF"}
        what = ", ".join(f"{names.get(r, r)} after {i} iterations" for r, i in zip(self.reasons, self.iterations) if r <= 0)
        super().__init__(f"KSPSolver[{prefix}]: KSPSolve has not converged ({what})")


def _truthy(v) -> bool:
    return v not in (False, 0, "false", "0", "False", None, "no", "off")


def _env_int(name: str, default: int) -> int:
    """Tuning defaults from the environment, read by the HOST layer once per solver and handed to the library per
    call (the library itself keeps no process-wide switch for them)."""
    try:
        return int(os.environ[name])
    except (KeyError, ValueError):
        return default

DIRECT_RTOL = 1e-12
CG_MERGED_MAX_ROWS = 1 << 20
KRYLOV_TYPES = ("cg", "bcgs", "bicgstab", "ibcgs", "pipebcgs", "fbcgsr")
HONOURED_KEYS = ("ksp_type", "pc_type", "ksp_rtol", "ksp_atol", "ksp_divtol", "ksp_max_it", "ksp_initial_guess_nonzero",
                 "ksp_error_if_not_converged", "ksp_cg_single_reduction", "ksp_cg_merged_reduction",
                 "ksp_bcgs_merged_reduction", "ksp_bcgs_restarts", "ksp_cg_fold_blocks", "ksp_run_ahead")
# keys the reference itself sets next to a direct solver (fracstep.py:565-570): they configure MUMPS, which the Krylov
# stand-in has no use for -- accepted silently with ksp_type=preonly
DIRECT_ONLY_KEYS = ("pc_factor_mat_solver_type", "mat_mumps_icntl_24", "mat_mumps_icntl_25")


class KSPSolver:
    def __init__(self, comm=None, petsc_options: typing.Optional[dict] = None, prefix="oasis_solver"):
        self._comm = comm
        self._prefix = prefix
        self._options: dict = {}
        self._A: SellMatrix | None = None
        self._dinv = None
        self._dinv_version = -1
        self._work = None
        self.last_result = None
        self.check_every = None  # override of the automatic check interval (see solve_block)
        self._every = {}
        self._env_fold = _env_int("OX_CG_FOLD_BLOCKS", -1)
        self._env_ahead = _env_int("OX_KSP_RUN_AHEAD", -1)
        self.updateOptions({} if petsc_options is None else petsc_options)

    # -- reference surface --------------------------------------------------------------
    def updateOptions(self, options: dict):
        """Update options (reference ksp.py:38-53)."""
        self._options.update({str(k): v for k, v in options.items()})

    def _audit_options(self):
        """Warn -- once per solver and (key, value) -- about every option that is remapped or ignored."""
        log = logging.getLogger("oasisx")
        seen = self.__dict__.setdefault("_warned", set())

        def warn(key, msg, level=logging.WARNING):
            tag = (key, str(self._options.get(key)))
            if tag not in seen:
                seen.add(tag)
                log.log(level, "KSPSolver[%s]: %s", self._prefix, msg)

        o = self._options
        kt = str(o.get("ksp_type", "")).lower()
        pc = str(o.get("pc_type", "")).lower()
        sym = self._A is not None and self._A.symmetric
        stand_in = "cg" if sym else "bcgs"
        direct = kt == "preonly"
        if direct:
            warn("ksp_type", f"ksp_type=preonly (pc_type={pc or 'unset'}): no sparse direct solver on the device; runs "
                             f"{stand_in}+jacobi to rtol {DIRECT_RTOL:g} and reports KSP_CONVERGED_ITS")
        elif kt == "":
            warn("ksp_type", f"ksp_type unset (PETSc's default: gmres): runs {stand_in}", logging.INFO)
        elif kt not in KRYLOV_TYPES:
            warn("ksp_type", f"ksp_type={kt} is not available on the device: runs {stand_in}")
        if pc == "":
            if not direct:
                warn("pc_type", "pc_type unset (PETSc's default: ilu / bjacobi): runs jacobi", logging.INFO)
        elif pc not in ("jacobi", "none") and not (direct and pc in ("lu", "cholesky")):
            warn("pc_type", f"pc_type={pc} is not available on the device: runs jacobi")
        elif pc == "none" and direct:
            warn("pc_type", "pc_type=none with ksp_type=preonly solves nothing in PETSc; runs the Krylov stand-in")
        for k in o:
            if k in HONOURED_KEYS or (direct and k in DIRECT_ONLY_KEYS):
                continue
            warn(k, f"option {k}={o[k]!r} is not known to this path and is ignored")

    def _pc_none(self) -> bool:
        return str(self._options.get("pc_type", "")).lower() == "none" and \
            str(self._options.get("ksp_type", "")).lower() != "preonly"

    def setOptions(self, op):
        """Reference ksp.py:55-59 gives the operator the solver's options prefix; nothing to
        do here -- matrix options do not exist on this path."""
        return None

    def setOperators(self, A: SellMatrix, P: typing.Optional[SellMatrix] = None):
        self._A = A
        self._dinv_version = -1
        self._every = {}
        self._merged_auto = None  # decided at the first one-column CG solve on this operator (see _cg_merged)

    def solve(self, b, x: Function) -> int:
        """Solve A x = b for one scalar field (reference ksp.py:71-78); returns the
        KSPConvergedReason.  ``b`` is a Vector (``Function.x.petsc_vec``) or a Function."""
        bvec = b.x if isinstance(b, Function) else b
        bs, bc = bvec._s, bvec._c
        xs, xc = x._storage, x._comp
        if bs.nc == 1 and xs.nc == 1:
            return int(self.solve_block(bs, xs)[0])
        # one column of an interleaved block: gather, solve, scatter back
        n = xs.n
        tb = FieldStorage(n, 1, xs.dev().device, n_alloc=xs.n_alloc)
        tx = FieldStorage(n, 1, xs.dev().device, n_alloc=xs.n_alloc)
        tb.dev()[:, 0].copy_(bs.dev()[:, 0 if bc is None else bc])
        tx.dev()[:, 0].copy_(xs.dev()[:, xc])
        reason = self.solve_block(tb, tx)[0]
        xs.dev()[:, xc].copy_(tx.dev()[:, 0])
        xs.mark_written()
        return int(reason)

    # -- block interface used by FractionalStep_AB_CN -------------------------------------
    def _method(self):
        o = self._options
        kt = str(o.get("ksp_type", "")).lower()
        direct = kt == "preonly"
        # "cg" on a mesh-partitioned operator runs the single-reduction recurrences (PETSc's
        # -ksp_cg_single_reduction, Chronopoulos-Gear: ONE merged reduction = one all-reduce per iteration,
        # 3 kernels instead of 5); same Krylov space, same convergence test, iteration counts within +-2 of
        # the standard recurrences (rounding only).  On ONE GPU the standard recurrences stay the default:
        # measured at 128^3 the pressure iteration takes 93.8 us single-reduction against 80.7 us standard
        # (one more vector: matrix + vectors no longer fit the 256 MB Infinity Cache between iterations).
        # The option forces either.
        dflt = self._A is not None and self._A.pattern.dist is not None
        single = o.get("ksp_cg_single_reduction", dflt) not in (False, 0, "false", "0", "False")
        cg = _lib.KSP_CG_SINGLE if single else _lib.KSP_CG
        # "bcgs" on a mesh-partitioned operator: the merged-reduction recurrences (2 all-reduces per iteration
        # instead of 3; same Krylov space, iteration counts within +-2)
        merged = o.get("ksp_bcgs_merged_reduction", dflt or kt in ("ibcgs", "pipebcgs", "fbcgsr")) \
            not in (False, 0, "false", "0", "False")
        bcgs = _lib.KSP_BCGS_MERGED if merged else _lib.KSP_BCGS
        if kt == "cg":
            meth = cg
        elif kt in ("bcgs", "bicgstab", "ibcgs", "pipebcgs", "fbcgsr"):
            meth = bcgs
        else:
            meth = cg if (self._A is not None and self._A.symmetric) else bcgs
        if direct:
            return meth, DIRECT_RTOL, 1e-50, 20000, True
        return (meth, float(o.get("ksp_rtol", 1e-5)), float(o.get("ksp_atol", 1e-50)),
                int(o.get("ksp_max_it", 10000)), False)

    def _cg_merged(self) -> bool:
        """One-column CG solves: the merged-reduction recurrences (OX_KSP_CG_MERGED)?  Default: on operators of at
        most CG_MERGED_MAX_ROWS (local) rows, where an iteration is bound by its launches and synchronisation points
        (tools/cg_iter_bench.py, P1 pressure matrix: 64^3 22.5 against 27.7 us per iteration, 128^3 70 against 66,
        256^3 482 against 432); the option forces either."""
        v = self._options.get("ksp_cg_merged_reduction")
        if v is not None:
            return v not in (False, 0, "false", "0", "False")
        if self._A is None:
            return False
        if getattr(self, "_merged_auto", None) is None:
            if self._A.pattern.dist is not None:
                # mesh-partitioned operators: ALWAYS.  One all-reduce per iteration like the single-reduction form, no
                # extra vectors (7 vector passes against its 11): on the self-loop plan of one rank (tools/
                # predict_scaling.py, round 5) 73 / 106 / 253 us per iteration at 1.1 / 2.1 / 8.5 M local rows against
                # 73 / 107 / 291 for the Chronopoulos-Gear recurrences.  (The same on every rank: no agreement needed.)
                self._merged_auto = True
            else:
                self._merged_auto = self._A.pattern.n_rows <= CG_MERGED_MAX_ROWS
        return self._merged_auto

    def _fold_blocks(self) -> int:
        """Blocks of the folded one-column CG update kernels this solver asks for (-1: the library's default)."""
        v = self._options.get("ksp_cg_fold_blocks")
        return int(v) if v is not None else self._env_fold

    def _run_ahead(self) -> int:
        v = self._options.get("ksp_run_ahead")
        return (1 if _truthy(v) else 0) if v is not None else self._env_ahead

    def _cg_folded(self) -> bool:
        """One-column standard CG on one GPU: the iteration's two synchronisation points are folded into the update
        kernels (3 kernels per iteration: csrc/ox_ksp.hip k_cg_update1f / k_cg_update2f; option ``ksp_cg_fold_blocks``).
        Partitioned operators never fold: their points carry an all-reduce."""
        if self._A is None or self._A.pattern.dist is not None:
            return False
        fb = self._fold_blocks()
        return (self._method()[0] == _lib.KSP_CG and not self._cg_merged()
                and (fb if fb >= 0 else int(_lib.load().ox_ksp_default_fold_blocks())) > 0)

    def _cg_kernels_per_iteration(self) -> int:
        """Kernels of one iteration of a one-column CG solve with this solver (reporting only: bench.py), as the LIBRARY
        decides it for this operator, method, check interval and fold setting (``ox_ksp_kernels_per_iteration``)."""
        if self._A is None:
            return 5
        meth = self._method()[0]
        if meth in (_lib.KSP_CG, _lib.KSP_CG_SINGLE) and self._cg_merged():
            meth = _lib.KSP_CG_MERGED
        every = self.check_every or self._every.get((1, meth)) or self._interval_for(1, meth)
        return int(_lib.load().ox_ksp_kernels_per_iteration(meth, self._A.ref(), 1, int(every), self._fold_blocks(),
                                                            int(self._A.pattern.dist is not None)))

    def solve_block(self, B: FieldStorage, X: FieldStorage, ax0: FieldStorage | None = None):
        """Solve A X = B for all ``nc`` interleaved right-hand sides in lockstep.
        Returns the list of per-component converged reasons.  ``ax0``: the product A X of the initial
        guess where the caller has it at hand (used with ``ksp_initial_guess_nonzero`` only): the
        solver's first mat-vec is skipped, same iterates."""
        if self._A is None:
            raise RuntimeError("KSPSolver.solve called before setOperators")
        lib = _lib.load()
        A = self._A
        nc = X.nc
        meth, rtol, atol, max_it, direct = self._method()
        if meth in (_lib.KSP_CG, _lib.KSP_CG_SINGLE) and nc == 1 and self._cg_merged():
            meth = _lib.KSP_CG_MERGED
        guess = bool(self._options.get("ksp_initial_guess_nonzero", False)) and not direct
        st = _lib.current_stream()
        dev = X.dev().device
        if self._dinv is None or self._dinv.shape[0] != A.pattern.n_rows:
            self._dinv = torch.empty(A.pattern.n_rows, dtype=torch.float64, device=dev)
            self._dinv_version = -1
        self._audit_options()
        pc_key = (A.version, self._pc_none())
        if self._dinv_version != pc_key:
            if pc_key[1]:  # pc_type none: the identity through the same kernels
                self._dinv.fill_(1.0)
            else:
                _lib.check(lib.ox_jacobi_setup(A.ref(), _lib.ptr(self._dinv), st), "ox_jacobi_setup")
            self._dinv_version = pc_key
            # a matrix that carries a value dictionary (la.SellMatrix.freeze: it will not change any more) has
            # few distinct diagonal values too: the CG update kernels then read one byte of dinv per row
            self._dcode = self._ddict = None
            if A.vcode is not None and dev.type == "cuda":
                code = torch.empty(A.pattern.n_rows, dtype=torch.uint8, device=dev)
                vdict = torch.zeros(256, dtype=torch.float64, device=dev)
                nd = C.c_int(0)
                _lib.check(lib.ox_value_dictionary(_lib.ptr(self._dinv), A.pattern.n_rows, 1, _lib.ptr(code), _lib.ptr(vdict),
                                                   C.byref(nd), st), "ox_value_dictionary")
                if nd.value > 0:
                    self._dcode, self._ddict = code, vdict[: nd.value]
        need = lib.ox_ksp_work_bytes_for(A.ref(), nc, meth)
        if self._work is None or self._work.shape[0] < need:
            self._work = torch.empty(int(need), dtype=torch.uint8, device=dev)
        res = _lib.ox_ksp_result()
        # Iterations enqueued between two host reads of the device state (a read costs ~30 us of
        # idle GPU).  One right-hand side: amortise the read over ~1.5 ms of iterations, at most 16
        # (pressure CG: 14-16).  Several in lockstep: a column that has converged should leave the
        # lockstep at once (narrowing) -- being k iterations late costs k half-price iterations --
        # so check every iteration unless iterations are shorter than two reads.
        key = (nc, meth)
        if key not in self._every:
            self._every[key] = self._interval_for(nc, meth)
        every = self.check_every or self._every[key]
        # a direct solver never breaks down: when one was asked for, let BiCGStab re-seed its shadow
        # residual on a rho/omega breakdown; an explicit "bcgs" behaves like PETSc's (reason -5)
        restarts = int(self._options.get("ksp_bcgs_restarts", 5 if direct else 0))
        dcode = getattr(self, "_dcode", None)
        opt = _lib.ox_ksp_options()
        _lib.check(lib.ox_ksp_options_default(C.byref(opt)), "ox_ksp_options_default")
        opt.rtol, opt.atol, opt.max_it = rtol, atol, max_it
        opt.divtol = float(self._options.get("ksp_divtol", 1e4))  # PETSc's default ("divergence=10000." in -ksp_view)
        opt.nonzero_guess, opt.check_every, opt.max_restarts = int(guess), int(every), restarts
        opt.fold_blocks, opt.run_ahead = self._fold_blocks(), self._run_ahead()
        opt.ax0 = ax0.ptr() if (ax0 is not None and guess) else None
        if dcode is not None:
            opt.dinv_code, opt.dinv_dict = _lib.ptr(dcode), _lib.ptr(self._ddict)
            opt.n_dinv_dict = int(self._ddict.shape[0])
        _lib.check(lib.ox_ksp_solve_opt(meth, A.ref(), _lib.ptr(self._dinv), B.ptr(), X.ptr(), nc, C.byref(opt),
                                        _lib.ptr(self._work), int(self._work.shape[0]), C.byref(res), A.pattern.dist, st),
                   "ox_ksp_solve")
        if A.pattern.dist is not None:  # x.scatter_forward() (reference ksp.py:77)
            _lib.check(lib.ox_halo_forward(A.pattern.dist, X.ptr(), nc, st), "ox_halo_forward")
        self.last_result = res
        reasons = [int(res.reason[c]) for c in range(nc)]
        if direct:
            reasons = [_lib.CONVERGED_ITS if r > 0 else r for r in reasons]
        if _truthy(self._options.get("ksp_error_if_not_converged", False)) and any(r <= 0 for r in reasons):
            raise KSPConvergenceError(self._prefix, reasons, [int(res.its[c]) for c in range(nc)])
        return reasons

    def _interval_for(self, nc: int, meth: int) -> int:
        # from the matrix size alone (bytes per iteration at ~4 TB/s + launch latencies), NOT from a measured time:
        # with several columns the interval decides at which iteration a solve narrows to its last live column,
        # the 1-column kernels sum their dot products in another order than the NC-column ones, and a schedule
        # that followed timing noise made the last bits of a step differ from run to run (round 4: seen as 1e-15
        # differences between identical runs at sizes where an iteration takes about the 60 us of the threshold)
        t_iter = (2 if meth in (_lib.KSP_BCGS, _lib.KSP_BCGS_MERGED) else 1) * 10.0 * self._A.pattern.nnz / 4.0e12 + 25e-6
        return self._check_interval(nc, t_iter)

    def _check_interval(self, nc: int, t_iter: float) -> int:
        """Iterations enqueued between two host reads of the device state, from the time of one iteration."""
        if nc > 1:
            ev = 1 if t_iter > 60e-6 else 4
        else:
            ev = max(1, min(16, int(1.5e-3 / t_iter)))
        A = self._A
        if A is not None and A.pattern.dist is not None and getattr(self._comm, "size", 1) > 1:
            # every rank must enqueue the same number of iterations (each carries exchanges)
            ev = int(self._comm.allreduce(ev, op="max"))
        return ev

    @property
    def iterations(self):
        r = self.last_result
        return [] if r is None else [int(v) for v in r.its]
