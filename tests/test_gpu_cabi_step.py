"""GPU: one IPCS time step driven through the C ABI ALONE -- demo/cabi_ipcs_step.py: numpy arrays,
ctypes and include/oasisx_hip.h; neither the oasisx_amd package (no fem.py) nor torch is imported in
that process -- against the oracle on the same mesh with the oracle's own dof numbering (fields
matched through the dof coordinates).  The mesh generator of the driver (alternating diagonals /
5 tetrahedra per cube) is neither the product's nor the oracle's."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("dim,N,udeg,compress,merged,windows", [
    (2, 12, 2, False, False, False), (3, 5, 2, False, True, False), (3, 6, 1, False, False, False), (3, 6, 2, True, True, False),
    (2, 16, 1, True, False, False), (3, 8, 2, True, False, True), (2, 24, 2, False, False, True)])
def test_ipcs_step_through_ctypes_only(hip, tmp_path, dim, N, udeg, compress, merged, windows):
    from oracle import ipcs_oracle as O
    from oracle.cpu_baseline import match_by_coordinates
    from tests.helpers import KRYLOV

    out = str(tmp_path / "step.npz")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "demo", "cabi_ipcs_step.py"), "--dim", str(dim), "-N", str(N),
                        "--udeg", str(udeg), "--steps", "2", "--out", out] + (["--compress"] if compress else [])
                       + (["--cg-merged"] if merged else [])  # (OX_KSP_CG_MERGED for the pressure solve)
                       + (["--windows"] if windows else []),  # brick order + ox_space_windows / ox_window_retile
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    g = np.load(out)
    if compress:  # ox_value_dictionary and ox_pair_stream_size/_fill were driven through the binding
        built = eval(str(g["compressed"]))
        # (the P2 matrices of this two-shape mesh have more than 256 distinct values: declined, as designed)
        assert built["Ap"]["n_dict"] > 0 and built["Ap"]["pair_codes"] > 0 and set(built) == {"M", "K", "Ap"}, built
    assert not bool(g["imported_package"]) and not bool(g["imported_torch"])  # the C ABI was all it used
    if windows:  # the velocity matrices really carried window blocks
        assert int(g["window_blocks"]) > 0 and int(g["window_max"]) > 0
    nu, dt = 0.01, 0.005
    R, clock = O.taylor_green_problem(0, dim, u_deg=udeg, p_deg=1, nu=nu, dt=dt, solver_options=KRYLOV,
                                      mesh=(g["coords"], g["cells"].astype(np.int64)))
    t = 0.0
    for _ in range(2):
        t += dt
        clock["t"] = t
        R.solve(dt, nu, max_iter=1)
    lo, hi = -np.ones(dim), np.ones(dim)
    pv = match_by_coordinates(g["x_v"], R.F.x_v, lo, hi)
    pq = match_by_coordinates(g["x_q"], R.F.x_q, lo, hi)
    assert np.abs(g["u"][pv] - R.u1).max() < 1e-8
    assert np.abs(g["p"][pq] - R.p).max() < 1e-7
    assert abs(int(g["its_pressure"][0]) - int(np.max(np.atleast_1d(R.its["pressure"])))) <= 2


@pytest.mark.parametrize("dim,N,deg,parts,split", [(3, 5, 2, 3, "slabs"), (2, 12, 2, 2, "slabs"), (3, 6, 1, 4, "slabs"),
                                                   (3, 8, 2, 8, "slabs"), (3, 6, 2, 8, "octants"), (3, 8, 1, 8, "octants")])
def test_partitioned_space_through_ctypes_only(hip, dim, N, deg, parts, split):
    """ox_mesh_create_sub / ox_space_create_part driven with numpy + ctypes only (demo/cabi_partitioned_space.py, its
    own numpy partition): every rank's owned mass-matrix rows equal the whole mesh's through the dof coordinates, the
    owned dofs of the ranks tile the space, ghosts follow the owned dofs ordered by (owner, initial id)."""
    import json

    r = subprocess.run([sys.executable, os.path.join(ROOT, "demo", "cabi_partitioned_space.py"), "--dim", str(dim), "-N", str(N),
                        "--degree", str(deg), "--parts", str(parts), "--split", split], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert not rep["imported_package"] and not rep["imported_torch"]
    assert len(rep["ranks"]) == parts and sum(q["owned"] for q in rep["ranks"]) == rep["n_global"]
    assert all(q["ghosts"] > 0 and q["max_rel_diff_of_owned_mass_rows"] < 1e-12 for q in rep["ranks"])


@pytest.mark.parametrize("dim,N,parts", [(3, 5, 2), (2, 12, 3)])
def test_partitioned_time_steps_through_ctypes_only(hip, dim, N, parts):
    """demo/cabi_partitioned_step.py: `parts` rank PROCESSES (sharing this GPU), each driving its part of two IPCS time
    steps with numpy + ctypes only -- sub-mesh, partitioned P2 / P1 spaces, halo plans on a caller-supplied transport
    (ox_dist_create_custom over pipes), operators, the three solves with their exchanges and all-reduces -- against the
    same steps on one rank: owned values and ghost copies agree to the solver tolerance, iteration counts within 2."""
    import json

    r = subprocess.run([sys.executable, os.path.join(ROOT, "demo", "cabi_partitioned_step.py"), "--dim", str(dim), "-N", str(N),
                        "--parts", str(parts), "--steps", "2"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert len(rep["ranks"]) == parts
    for q in rep["ranks"]:
        assert not q["imported_package"] and not q["imported_torch"]
        assert q["ghost_u"] > 0 and q["ghost_p"] > 0 and q["halo_exchanges"] > 10 and q["all_reduces"] > 10
        for step, ref in zip(q["iterations"], rep["reference_iterations"]):
            for name in ("tentative", "pressure", "update"):
                assert max(abs(a - b) for a, b in zip(step[name], ref[name])) <= 2, (name, step, ref)
    assert rep["max_rel_diff_u"] < 1e-8 and rep["max_rel_diff_p"] < 1e-7, rep
    assert rep["max_rel_diff_ghost_u"] < 1e-8 and rep["max_rel_diff_ghost_p"] < 1e-7, rep
