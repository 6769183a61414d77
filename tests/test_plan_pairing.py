"""The partition shapes an 8-GPU run produces, built for ALL ranks in one process and paired against each other.

Every rank derives its halo plan from its own window of the replicated mesh, with no communication at set-up
(oasisx_amd/fem.py ``_build_halo``): rank r's send list to q and q's receive block from r are computed on two different
ranks and must name the same dofs in the same order.  A real job finds a mismatch in ``check_halo`` at set-up; here the
2 x 2 x 2 split of a box (7 peers per rank, among them peers that share one face, one P2 edge line or a single vertex
dof) and an 8-way split of a Delaunay mesh are checked pair by pair through the dof coordinates, which are the global
ids of Lagrange nodes -- what DOLFINx's index maps and ``scatter_forward`` give the reference on a distributed mesh
(reference fracstep.py:186-216,453,497,551,632,655).

CPU: the torch twin of the set-up (degree 1, 2).  GPU: the library's own set-up (``ox_mesh_create_sub`` /
``ox_space_create_part``), degrees 1-3 (the owned operator rows of 8 parts against the whole mesh's:
tests/test_gpu_cabi_step.py, demo/cabi_partitioned_space.py --parts 8 --split octants).
"""
import numpy as np
import pytest
import torch


def _mesh(kind, N, device):
    from oasisx_amd import mesh as M

    if kind == "box":
        return M.create_box(None, [[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]], [N, N, N], device=device)
    return M.create_delaunay_box(None, [[-1.0] * 3, [1.0] * 3], N, seed=4, device=device)


def _q(x):
    """Coordinates on a 2^-36 grid of [-1, 1]: equal for the same Lagrange node whichever rank computed it."""
    return np.round((np.asarray(x, dtype=np.float64) + 1.0) * float(1 << 35)).astype(np.int64)


def build_ranks(kind, N, degree, nparts, device):
    from oasisx_amd import fem
    from oasisx_amd.parallel import MeshPartition

    m = _mesh(kind, N, device)
    spaces = []
    for r in range(nparts):
        part = MeshPartition(m, r, nparts, faces=(degree == 3))
        spaces.append(fem.FunctionSpace(m, degree, window=128, part=part))
    return m, spaces


def check_pairing(m, spaces, degree):
    """Owned dofs tile the space; for every ordered pair (r, q) the dofs r sends to q are the dofs of q's receive
    block from r, in order; a pair exchanges in one direction exactly when the other side expects it."""
    P = len(spaces)
    X = [_q(V.x.cpu().numpy()) for V in spaces]
    owned = np.concatenate([X[r][: spaces[r].n_owned] for r in range(P)])
    uniq = np.unique(owned, axis=0)
    assert uniq.shape[0] == owned.shape[0] == spaces[0].num_dofs_global, (uniq.shape, owned.shape, spaces[0].num_dofs_global)
    plans = []
    for V in spaces:
        h = V.halo
        send_idx = h["send_idx"].cpu().numpy()
        plans.append({int(q): (send_idx[h["send_off"][i]: h["send_off"][i + 1]],
                               (int(h["recv_off"][i]), int(h["recv_off"][i + 1])))
                      for i, q in enumerate(h["peers"])})
        assert list(h["peers"]) == sorted(h["peers"]) and int(h["recv_off"][-1]) == V.n_local - V.n_owned
    counts = np.zeros((P, P), dtype=np.int64)
    for r in range(P):
        for q in range(P):
            if q == r:
                assert r not in plans[r]
                continue
            send = plans[r].get(q, (np.zeros(0, dtype=np.int32), (0, 0)))[0]
            r0, r1 = plans[q].get(r, (None, (0, 0)))[1]
            assert send.shape[0] == r1 - r0, f"rank {r} sends {send.shape[0]} dofs to {q}, which expects {r1 - r0}"
            if send.shape[0]:
                assert int(send.max()) < spaces[r].n_owned  # only owned values travel
                got = X[q][spaces[q].n_owned + r0: spaces[q].n_owned + r1]
                assert np.array_equal(X[r][send], got), f"P{degree}: the send list {r} -> {q} and the receive block disagree"
            counts[r, q] = send.shape[0]
            # grouped ncclSend / ncclRecv pair up only when both sides list each other
            assert (q in plans[r]) == (r in plans[q]), (r, q)
    return counts


def check_box_2x2x2(counts, degree, N):
    """The shapes slab splits never reach.  A rank keeps one layer of ghost cells around the dofs it owns, so it
    receives the other dofs of those cells from their owners: octants 0 and 7 exchange with all 7 others, and among the
    lists are planes of dofs (face neighbours), single lines (octants that meet in an edge of the split) and ONE dof
    (0 -> 7: the octants that meet in the centre vertex only)."""
    P = counts.shape[0]
    assert P == 8
    either = (counts + counts.T) > 0
    npeers = either.sum(axis=1)
    assert npeers[0] == 7 and npeers[7] == 7 and npeers.min() >= 3, npeers
    assert (counts[0] > 0).sum() == 7  # rank 0 owns every interface dof it touches: it sends to all seven
    assert (counts[:, 7] > 0).sum() == 7  # rank 7 owns none of them: it receives from all seven
    assert counts[0, 7] == 1  # the centre vertex
    k = degree * (N // 2) + 1  # Lagrange nodes along half an edge of the box, ends included
    pos = counts[counts > 0]
    assert ((pos > 1) & (pos <= k)).any()  # a pair that shares one line of dofs
    assert (pos > k).any()  # face neighbours: (half-)planes of dofs


@pytest.mark.parametrize("kind,N,degree", [("box", 8, 1), ("box", 8, 2), ("box", 6, 2), ("delaunay", 7, 1), ("delaunay", 7, 2)])
def test_eight_ranks_plans_pair_up_cpu_setup(kind, N, degree):
    m, spaces = build_ranks(kind, N, degree, 8, "cpu")
    counts = check_pairing(m, spaces, degree)
    if kind == "box":
        check_box_2x2x2(counts, degree, N)
    else:
        assert ((counts + counts.T) > 0).sum(axis=1).min() >= 3 and counts.sum() == sum(V.n_local - V.n_owned for V in spaces)


def test_p1_list_empty_where_the_p2_list_is_not():
    """Taylor-Hood on 8 ranks: between some octant pairs the P2 plan moves edge-midpoint dofs while the P1 plan of the same
    pair moves fewer (a single vertex, or none on an odd split) -- both spaces' plans pair up on the same partition."""
    from oasisx_amd import fem
    from oasisx_amd.parallel import MeshPartition

    m = _mesh("box", 5, "cpu")  # odd: the cuts run through cells, the octants' interfaces are staircases
    c = {}
    for degree in (1, 2):
        spaces = [fem.FunctionSpace(m, degree, window=128, part=MeshPartition(m, r, 8)) for r in range(8)]
        c[degree] = check_pairing(m, spaces, degree)
    assert ((c[2] > 0) & (c[1] == 0)).any()
    assert ((c[1] > 0) <= (c[2] > 0)).all()  # whoever exchanges vertices exchanges edge dofs too


@pytest.mark.gpu
@pytest.mark.parametrize("kind,N,degree", [("box", 8, 1), ("box", 8, 2), ("box", 6, 3), ("delaunay", 7, 2), ("delaunay", 6, 1)])
def test_eight_ranks_plans_pair_up_native_setup(hip, kind, N, degree):
    """The same through the library's set-up on the device (ox_mesh_create_sub / ox_space_create_part), P3 included."""
    m, spaces = build_ranks(kind, N, degree, 8, "cuda")
    counts = check_pairing(m, spaces, degree)
    if kind == "box":
        check_box_2x2x2(counts, degree, N)
    assert counts.sum() == sum(V.n_local - V.n_owned for V in spaces)
