"""SpMV micro-benchmark on the bench workload's matrices (x_j = sin(j*1e-3)+1,
200 repetitions after 20 warm-up), HIP-event timed.  VARIANTS=1,3 lists the kernel variants to compare
(bit 0 nontemporal matrix stream, bit 1 16-bit column stream, bit 2 1-byte value codes; PALETTE=49
draws the values from 49 distinct numbers so that a dictionary exists)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oasisx_amd import fem, _lib, mesh as M
from oasisx_amd.la import SellMatrix
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
which = sys.argv[2] if len(sys.argv) > 2 else "p"
mesh = M.create_box(None, [[-1.,-1.,-1.],[1.,1.,1.]], [N,N,N])
deg, nc = {"p": (1, 1), "u": (2, 3), "u1": (2, 1)}[which]  # u1: one column on the velocity matrix (narrowed solves)
V = fem.FunctionSpace(mesh, deg, window=int(os.environ.get("WINDOW", "4096")), brick=os.environ.get("BRICK") == "1")
if os.environ.get("WIN") == "1":  # LDS-window stream of the pattern (variant bit 4): VARIANTS=15,31 compares
    print("windows:", V.build_windows(), getattr(V.pattern, "w_stats", None))
A = SellMatrix(V.pattern); A.vals.uniform_(0.5, 1.5)
real = os.environ.get("REAL", "")  # "stiff" / "mass": the mesh's real stiffness / mass matrix, frozen
if real:
    import ctypes as C
    from oasisx_amd.fem import cell_geometry
    lib0 = _lib.load()
    geom = V.native.nmesh.geom if getattr(V, "native", None) is not None else cell_geometry(mesh, V.local_cells)
    cells = _lib.ox_cells(mesh.gdim, 0, int(geom.shape[0]), geom.data_ptr())
    adj = V.adj.struct()
    nb, bptr, bsl, bw = V.pattern.bins_args()
    _lib.check(lib0.ox_assemble_matrix({"mass": 0, "stiff": 1}[real], V.degree, C.byref(cells), _lib.ptr(V.cell_dofs),
                                       C.byref(adj), _lib.ptr(V.adj.adj_pos), V.adj.pw, A.ref(), nb, bptr, bsl, bw,
                                       _lib.current_stream()), "ox_assemble_matrix")
    A.version += 1
    print("value dictionary built:", A.freeze(pairs=os.environ.get("PAIRS", "auto")), "entries", A._struct.n_dict)
npal = int(os.environ.get("PALETTE", "0"))  # > 0: values drawn from that many distinct numbers (mass /
if npal:                                   # stiffness matrices on box meshes have 49 / 14)
    pal = torch.rand(npal, device="cuda", dtype=torch.float64) + 0.5
    A.vals.copy_(pal[torch.randint(0, npal, (A.vals.numel(),), device="cuda")])
    print("value dictionary built:", A.freeze(pairs=os.environ.get("PAIRS", "auto")), "entries", A._struct.n_dict)
P = V.pattern
x = (torch.sin(torch.arange(P.n_cols*nc, device="cuda", dtype=torch.float64)*1e-3)+1).reshape(P.n_cols, nc).contiguous()
y = torch.zeros_like(x)
lib = _lib.load()
B = 12*P.nnz + 4*(P.n_rows+1) + nc*8*(P.n_cols+P.n_rows)
variants = [int(v) for v in os.environ.get("VARIANTS", "1,3,7,15" if (npal or real) else "1,3").split(",")]
stored = P.size*((1 if A.vcode is not None else 8) + 2*P.frac16 + 4*(1-P.frac16)) + 8*(P.size//128) + nc*8*(P.n_cols+P.n_rows)
print(f"16-bit column stream covers {P.frac16:.4f} of the stored entries; CSR bytes {B/1e6:.1f} MB, stored bytes {stored/1e6:.1f} MB")
res = {v: [] for v in variants}
ref_y = None
for v in variants:  # every variant must reproduce the first one bit for bit
    A.set_levels(v); y.zero_(); A.mult(x, y, nc); torch.cuda.synchronize()
    if ref_y is None: ref_y = y.clone()
    else: print(f"variant {v} bit-identical to variant {variants[0]}: {torch.equal(ref_y, y)}")
for rnd in range(int(os.environ.get('ROUNDS', '7'))):
    for v in variants:
        A.set_levels(v)
        for _ in range(5): A.mult(x, y, nc)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = int(os.environ.get('REPS', '100'))
        e0.record()
        for _ in range(reps): A.mult(x, y, nc)
        e1.record(); torch.cuda.synchronize()
        res[v].append(e0.elapsed_time(e1)*1e3/reps)
import statistics
for v in variants:
    med, mn = statistics.median(res[v]), min(res[v])
    if (v & 30) == 30 and nc == 1 and A.pw_code is not None:  # pair-window stream: 4 B per slot + 4 B per distinct pair + vectors
        sb = 4*A.pw_code.numel() + 4*A.pw_list.numel() + 8*A.pw_ptr.numel() + 8*(P.n_slices+1) + 8*(P.n_cols+P.n_rows)
        print(f"pair-window stream: {A._struct.pw_slices} slices per block, {A.pw_list.numel()/P.n_rows:.2f} pairs loaded per row, largest window {A._struct.pw_max} pairs, {sb/1e6:.1f} MB moved")
    elif (v & 14) == 14 and A.ps_code is not None:  # pair-slot stream: 4 B per slot + bases + vectors
        sb = 4*A.ps_code.numel() + 8*(A.ps_code.numel()//256) + 8*(P.n_slices+1) + nc*8*(P.n_cols+P.n_rows)
        print(f"pair-slot stream: {A.ps_code.numel()/64/max(P.n_slices,1):.1f} slots per row stored for {P.nnz/P.n_rows:.1f} entries, {A.ps_wide} wide slices, {sb/1e6:.1f} MB moved")
    else: sb = stored if (v & 6) == 6 else (B if not (v & 2) else stored + (7*P.size if A.vcode is not None else 0))
    print(f"variant={v} {which} N={N} median={med:.1f} us min={mn:.1f} us | CSR-priced {B/med/1e3:.0f} GB/s | bytes moved {sb/med/1e3:.0f} GB/s = {sb/med/1e3/8000:.3f} of 8 TB/s")
