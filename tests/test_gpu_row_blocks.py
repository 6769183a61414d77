"""GPU: the one-launch row-block form of the row kernels (``ox_assemble_first_blocks`` / ``ox_assemble_matrix_blocks``,
round 5; reference fracstep.py:373-380,432-469 -> ``assemble_matrix`` + ``Mat`` passes) against the width-bin launches:
the same per-slice arithmetic in the same order, so every matrix value, b_first and the ``A u1`` by-product are
BIT-identical -- box meshes (three widths), Delaunay meshes (dozens), 2-D, P1, P3, mesh-partitioned row spaces; and the
block table of the library against its torch twin."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mesh(kind, dim, n):
    from oasisx_amd import mesh as M

    if kind == "box":
        return (M.create_rectangle(None, [[-1.0, -1.0], [1.0, 1.0]], [n, n + 1]) if dim == 2 else
                M.create_box(None, [[-1.0] * 3, [1.0] * 3], [n, n + 1, n - 1]))
    return M.create_delaunay_box(None, [[-1.0] * dim, [1.0] * dim], n, refine=1)


@pytest.mark.parametrize("kind,dim,n,udeg,window,dictionary", [
    ("box", 3, 8, 2, 4096, True), ("box", 3, 9, 2, 256, False), ("box", 2, 24, 2, 4096, True), ("box", 3, 10, 1, 4096, True),
    ("delaunay", 3, 6, 2, 4096, False), ("delaunay", 2, 14, 2, 1024, False), ("box", 2, 10, 3, 4096, False)])
def test_row_blocks_reproduce_the_width_bins_bit_for_bit(hip, kind, dim, n, udeg, window, dictionary):
    import oasisx_amd as ox
    from oasisx_amd import fem
    from tests.helpers import KRYLOV

    mesh = _mesh(kind, dim, n)
    got = {}
    for blocks in (False, True):
        bcs = [[ox.DirichletBC(0.0, ox.LocatorMethod.GEOMETRICAL, lambda x: np.isclose(np.abs(x[0]), 1.0))] for _ in range(dim)]
        so = {k: dict(v, ksp_initial_guess_nonzero=True) for k, v in KRYLOV.items()}
        S = ox.FractionalStep_AB_CN(mesh, ("Lagrange", udeg), ("Lagrange", 1 if udeg < 3 else 2), bcs_u=bcs, bcs_p=[],
                                    solver_options=so, options={"sell_window": window, "assemble_row_blocks": blocks,
                                                                "value_dictionary": dictionary})
        P = S._A.pattern
        assert S._row_blocks == blocks and P.n_row_blocks > 0
        g = torch.Generator(device="cuda").manual_seed(7)
        for F in (S._U1, S._U2, S._B0):
            F.dev().copy_(torch.randn(F.dev().shape, dtype=torch.float64, device="cuda", generator=g))
        S.assemble_first(0.01, 0.02)
        got[blocks] = [t.clone() for t in (S._M.vals, S._K.vals, S._Ap.vals, S._A.vals, S._BFIRST.dev(), S._B3.dev())]
        if blocks:
            # the library's block table = the torch twin's; blocks are consecutive, within the two budgets, and cover every slice
            bp = P.row_blk_ptr.cpu().numpy()
            tw, big = fem.row_blocks(P.widths)
            assert np.array_equal(bp, tw) and big == P.row_blk_entries
            assert bp[0] == 0 and bp[-1] == P.n_slices and (np.diff(bp) >= 1).all() and (np.diff(bp) <= 8).all()
            ent = np.add.reduceat(P.widths.astype(np.int64) * 64, bp[:-1])
            assert ent.max() == big and big * 8 <= 134 * 1024
    names = ("M", "K", "Ap", "A", "b_first", "A u1")
    for name, a, b in zip(names, got[False], got[True]):
        assert torch.equal(a, b), name
    assert float(got[True][3].abs().sum()) > 0 and float(got[True][5].abs().sum()) > 0


def test_row_blocks_fall_back_to_the_bins_for_rows_beyond_the_lds_budget(hip):
    """A slice wider than OX_ROW_BLOCK_LDS / 512 entries has no row block: the pattern reports none and the solver runs the
    width bins (which size their launch for that one width, up to 140 KB)."""
    from oasisx_amd import fem

    w = np.array([30, 30, 269, 20], dtype=np.int32)
    bp, big = fem.row_blocks(w)
    assert bp.tolist() == [0] and big == 0
    bp, big = fem.row_blocks(np.array([65] * 9 + [27] * 9 + [19] * 3, dtype=np.int32))
    # 4 x 65 fit (133 120 B); 65 + 7 x 27 fit too but 8 waves cap the block; 27 + 27 + 3 x 19 is the rest
    assert bp.tolist() == [0, 4, 8, 16, 21] and big == 4 * 65 * 64
    assert fem.row_blocks(np.zeros(0, dtype=np.int32))[0].tolist() == [0]
