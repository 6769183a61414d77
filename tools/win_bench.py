"""LDS-window SpMV (k_spmv_win) against the lane = row kernels on the velocity matrices of the bench workload:
the mass matrix through its value dictionary and an f64-valued matrix on the same pattern (what A is), one and three
right-hand sides, line order (OX_BRICK=0) and brick order of the P2 numbering.  Bit-identity of y is checked for
every pair; times are HIP-event medians, interleaved in one process.

    python tools/win_bench.py [N] [delaunay|box] [refine]"""
import ctypes as C
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oasisx_amd import _lib, fem
from oasisx_amd import mesh as M
from oasisx_amd.la import SellMatrix

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
kind = sys.argv[2] if len(sys.argv) > 2 else "box"
refine = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lib = _lib.load()
REPS = int(os.environ.get("REPS", "30"))
ROUNDS = int(os.environ.get("ROUNDS", "5"))
WINDOW = int(os.environ.get("WINDOW", "4096"))


def timed(A, x, y, nc, variant):
    A.set_levels(variant)
    for _ in range(3):
        A.mult(x, y, nc)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        A.mult(x, y, nc)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / REPS


for brick in ([int(b) for b in os.environ.get("BRICKS", "1,0").split(",")] if kind == "box" else [1]):
    os.environ["OX_BRICK"] = str(brick)
    if kind == "box":
        mesh = M.create_box(None, [[-1., -1., -1.], [1., 1., 1.]], [N, N, N])
    else:
        mesh = M.create_delaunay_box(None, [[-1., -1., -1.], [1., 1., 1.]], N, refine=refine)
    t0 = time.perf_counter()
    V = fem.FunctionSpace(mesh, 2, window=WINDOW)
    torch.cuda.synchronize()
    t_space = time.perf_counter() - t0
    P = V.pattern
    Mm = SellMatrix(P, symmetric=True, name="M")
    geom = V.native.nmesh.geom
    cells = _lib.ox_cells(mesh.gdim, 0, int(geom.shape[0]), geom.data_ptr())
    adj = V.adj.struct()
    nb, bptr, bsl, bw = P.bins_args()
    _lib.check(lib.ox_assemble_matrix(0, 2, C.byref(cells), _lib.ptr(V.cell_dofs), C.byref(adj), _lib.ptr(V.adj.adj_pos),
                                      V.adj.pw, Mm.ref(), nb, bptr, bsl, bw, _lib.current_stream()), "ox_assemble_matrix")
    Mm.version += 1
    Am = SellMatrix(P, name="A")
    Am.vals.copy_(Mm.vals * (1.0 + 0.25 * torch.sin(torch.arange(P.size, device="cuda", dtype=torch.float64))))
    Am.version += 1
    t0 = time.perf_counter()
    built = V.build_windows(int(os.environ.get("SPLIT", "0")))  # SPLIT=2176: over-budget blocks cut in two
    torch.cuda.synchronize()
    t_win = time.perf_counter() - t0
    # the structs were made before the windows existed: refresh them
    for A in (Mm, Am):
        S = P.struct(A.vals)
        for f in ("wb_slices", "wb_waves", "wb_ptr", "wlist", "wt_ptr", "wcode", "n_wblocks", "w_max"):
            setattr(A._struct, f, getattr(S, f))
    frozen = Mm.freeze(pairs="never")
    print(f"--- {kind} N={N} refine={refine} brick={brick}: rows {P.n_rows} nnz {P.nnz} slots {P.size} (padding "
          f"{100 * (P.size / P.nnz - 1):.2f} %), 16-bit columns {P.frac16:.3f}, dictionary {frozen}; space {t_space:.2f} s, "
          f"windows {t_win:.2f} s: {P.w_stats}", flush=True)
    for name, A in (("M (codes)" if frozen else "M (f64)", Mm), ("A (f64)", Am)):
        for nc in (1, 3):
            x = (torch.sin(torch.arange(P.n_cols * nc, device="cuda", dtype=torch.float64) * 1e-3) + 1).reshape(P.n_cols, nc).contiguous()
            y0, y1 = torch.zeros_like(x), torch.zeros_like(x)
            A.set_levels(15)
            A.mult(x, y0, nc)
            A.set_levels(31)
            A.mult(x, y1, nc)
            torch.cuda.synchronize()
            same = torch.equal(y0, y1)
            t = {15: [], 31: []}
            for _ in range(ROUNDS):
                for v in (15, 31):
                    t[v].append(timed(A, x, y1, nc, v))
            vb = 1 if (A.vcode is not None) else 8
            stream = P.size * (vb + 2 * P.frac16 + 4 * (1 - P.frac16)) + 8 * (P.size // 128) + nc * 8 * (P.n_cols + P.n_rows)
            wstream = P.wcode.numel() * (2 + (1 if vb == 1 else 0)) + (8 * P.size if vb == 8 else 0) + 4 * P.wlist.numel() \
                + nc * 8 * (P.wlist.numel() + P.n_rows)
            a, b = statistics.median(t[15]), statistics.median(t[31])
            print(f"{name:10s} nc={nc}: lane=row {a:8.1f} us ({stream / a / 1e3 / 8000:.3f} of peak on {stream / 1e6:.0f} MB) | window "
                  f"{b:8.1f} us ({wstream / b / 1e3 / 8000:.3f} on {wstream / 1e6:.0f} MB) | x{a / b:.2f} | bit-identical {same}", flush=True)
    del V, P, Mm, Am, mesh
    torch.cuda.empty_cache()
