"""CPU, world_size 2 over gloo: the mesh-partition logic of the N > 1 path (owned/ghost numbering,
halo plan, distributed SpMV and Jacobi-CG with the all-reduced dot products) against the serial
oracle.  The local operators come from the oracle here (this is a test of the HOST logic; the
device kernels are covered by the -m gpu tests, including a partitioned run on one GPU)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _halo_forward(V, x, rank):
    """scatter_forward over gloo with the plan the device path uses (peers / send_idx / offsets)."""
    h = V.halo
    reqs, bufs = [], []
    for i, q in enumerate(h["peers"]):
        s0, s1 = h["send_off"][i], h["send_off"][i + 1]
        r0, r1 = h["recv_off"][i], h["recv_off"][i + 1]
        if s1 > s0:
            sb = torch.from_numpy(x[h["send_idx"].numpy()[s0:s1]].copy())
            reqs.append(dist.isend(sb, int(q)))
            bufs.append(sb)
        if r1 > r0:
            rb = torch.empty(int(r1 - r0), dtype=torch.float64)
            reqs.append(dist.irecv(rb, int(q)))
            bufs.append((rb, r0, r1))
    for r in reqs:
        r.wait()
    for b in bufs:
        if isinstance(b, tuple):
            x[V.n_owned + b[1]:V.n_owned + b[2]] = b[0].numpy()


def _worker(rank, world, port, dim, N, deg, out, kind="box"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oasisx_amd import fem
        from oasisx_amd import mesh as M
        from oasisx_amd.parallel import MeshPartition, init_comm
        from oracle import ipcs_oracle as O

        comm = init_comm()
        assert comm.rank == rank and comm.size == world and comm.handle is None
        if kind == "delaunay":  # genuinely unstructured (irregular valence), any number of parts
            from tests.helpers import delaunay_box_mesh

            pts, tets = delaunay_box_mesh(N, dim, seed=1)
            m = M.from_arrays(pts, tets, comm=comm, device="cpu")
        else:
            m = (M.create_rectangle(comm, [[-1, -1], [1, 1]], [N, N], device="cpu") if dim == 2
                 else M.create_box(comm, [[-1, -1, -1], [1, 1, 1]], [N, N, N], device="cpu"))
        part = MeshPartition(m, rank, world)
        V = fem.FunctionSpace(m, deg, window=128, part=part)
        n_tot = torch.tensor([V.n_owned])
        dist.all_reduce(n_tot)
        assert int(n_tot) == V.num_dofs_global
        # local operator (owned rows) from the oracle on the local cells / local numbering
        lc = V.local_cells.numpy()
        F = O.Forms(m.coords.numpy(), m.cells.numpy()[lc], deg, 1, vd=V.cell_dofs.numpy(), qd=m.cells.numpy()[lc],
                    nv_dofs=V.n_local, nq_dofs=m.num_vertices)
        A = (F.stiffness_v() + 40.0 * F.mass_v())[: V.n_owned].tocsr()
        assert abs(V.pattern.to_csr(V.pattern.values_from_csr(A)) - A).max() == 0.0
        f = lambda x: np.sin(2 * x[:, 0]) + x[:, 1] ** 2 - 0.3 * x[:, -1]  # noqa: E731
        xs = V.x.numpy()
        # 1. halo: ghosts filled by the plan equal the function values at the ghost coordinates
        x = np.zeros(V.n_local)
        x[: V.n_owned] = f(xs[: V.n_owned])
        _halo_forward(V, x, rank)
        assert np.abs(x - f(xs)).max() < 1e-14
        # 2. distributed Jacobi-CG (PETSc conventions) == serial oracle CG on the global matrix
        b = np.cos(xs[: V.n_owned, 0]) * (1 + xs[: V.n_owned, 1])
        dinv = 1.0 / A.diagonal()

        def gsum(v):
            t = torch.tensor([float(v)], dtype=torch.float64)
            dist.all_reduce(t)
            return float(t)

        xl = np.zeros(V.n_local)
        r = b.copy()
        z = dinv * r
        p = np.zeros(V.n_local)
        p[: V.n_owned] = z
        rz = gsum(r @ z)
        bn = np.sqrt(gsum((dinv * b) @ (dinv * b)))
        its = 0
        while True:
            _halo_forward(V, p, rank)
            q = A @ p
            alpha = rz / gsum(p[: V.n_owned] @ q)
            xl[: V.n_owned] += alpha * p[: V.n_owned]
            r -= alpha * q
            z = dinv * r
            its += 1
            if np.sqrt(gsum(z @ z)) <= 1e-10 * bn:
                break
            rzn = gsum(r @ z)
            p[: V.n_owned] = z + (rzn / rz) * p[: V.n_owned]
            rz = rzn
        # serial reference on rank 0's copy of the global problem
        Vg = fem.FunctionSpace(m, deg, window=128)
        Fg = O.Forms(m.coords.numpy(), Vg.cells_in_kernel_order(), deg, 1, vd=Vg.cell_dofs.numpy(), qd=Vg.cells_in_kernel_order(),
                     nv_dofs=Vg.num_dofs, nq_dofs=m.num_vertices)
        Ag = (Fg.stiffness_v() + 40.0 * Fg.mass_v()).tocsr()
        xg = Vg.x.numpy()
        bg = np.cos(xg[:, 0]) * (1 + xg[:, 1])
        sol, reason, its_g, _ = O.jacobi_cg(Ag, bg, rtol=1e-10, atol=1e-50)

        def key(c):
            qq = np.round(c * 4096).astype(np.int64)
            k = qq[:, 0]
            for j in range(1, qq.shape[1]):
                k = k * (1 << 20) + qq[:, j]
            return k

        kg = key(xg)
        og = np.argsort(kg)
        idx = og[np.searchsorted(kg[og], key(xs[: V.n_owned]))]
        err = np.abs(xl[: V.n_owned] - sol[idx]).max()
        assert abs(its - its_g) <= 1, (its, its_g)
        assert err < 1e-9, err
        out[rank] = 1
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dim,N,deg,kind,world", [(2, 8, 2, "box", 2), (3, 4, 2, "box", 2), (3, 5, 1, "box", 2),
                                                  (3, 4, 2, "delaunay", 3), (2, 9, 2, "delaunay", 2)])
def test_partitioned_cg_over_gloo(dim, N, deg, kind, world):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), dim, N, deg, out, kind), nprocs=world, join=True)
    assert all(out.get(r) == 1 for r in range(world))


def test_recursive_coordinate_bisection_parts():
    from oasisx_amd.parallel import recursive_coordinate_bisection
    from tests.helpers import delaunay_box_mesh

    pts, tets = delaunay_box_mesh(6, 3, seed=2)
    cen = torch.from_numpy(pts[tets].mean(axis=1))
    for k in (2, 3, 5, 8):
        r = recursive_coordinate_bisection(cen, k)
        cnt = torch.bincount(r, minlength=k)
        assert int(cnt.max() - cnt.min()) <= 1 and int(cnt.sum()) == tets.shape[0]
        assert torch.equal(r, recursive_coordinate_bisection(cen, k))  # deterministic
    r8 = recursive_coordinate_bisection(cen, 8)  # compact: 2 x 2 x 2 octants of the cube
    for q in range(8):
        c = cen[r8 == q]
        assert float((c.max(dim=0).values - c.min(dim=0).values).max()) < 1.4


def test_mesh_files_round_trip(tmp_path):
    from oasisx_amd import mesh as M
    from tests.helpers import delaunay_box_mesh

    pts, tets = delaunay_box_mesh(3, 3, seed=3)
    m = M.from_arrays(pts, tets, device="cpu")
    for name in ("m.npz", "m.txt"):
        M.write_mesh(m, str(tmp_path / name))
        r = M.import_mesh(str(tmp_path / name)) if name.endswith("npz") else M.read_mesh(str(tmp_path / name), device="cpu")
        assert np.array_equal(r.cells.cpu().numpy(), tets) and np.abs(r.coords.cpu().numpy() - pts).max() == 0.0
    # Medit ASCII (1-based, reference tags), with an unused vertex that must be dropped
    with open(tmp_path / "m.mesh", "w") as f:
        f.write("MeshVersionFormatted 1\nDimension 3\nVertices %d\n" % (pts.shape[0] + 1))
        for p in pts:
            f.write("%.17g %.17g %.17g 0\n" % tuple(p))
        f.write("9 9 9 0\nTetrahedra %d\n" % tets.shape[0])
        for t in tets:
            f.write("%d %d %d %d 1\n" % tuple(t + 1))
        f.write("End\n")
    r = M.read_mesh(str(tmp_path / "m.mesh"), device="cpu")
    assert r.num_vertices == pts.shape[0] and np.array_equal(r.cells.cpu().numpy(), tets)
