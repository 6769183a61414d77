"""SpMV micro-benchmark on the bench workload's matrices (x_j = sin(j*1e-3)+1,
100 repetitions after 5 warm-up, 7 rounds interleaved), HIP-event timed.

    python tools/spmv_bench.py N which          which: p (P1, 1 column) | u (P2, 3) | u1 (P2, 1)

CONFIGS (env) lists what to compare, ';'-separated: "v<variant>" = the r01 kernel k_spmv (bit 0
nontemporal stream, bit 1 16-bit columns, bit 2 value codes), or "mode,waves,spw,wmax" = the packed
kernel k_spmv_pk (mode 1 packed stream, 2 + LDS x window, 9 diagnostic: every gather reads the row's own
x; spw > 0: slices per wave, spw = -k: persistent grid of k blocks per CU, slices dealt evenly).
PALETTE=n draws the values from n distinct numbers so that a value dictionary exists (mass /
stiffness matrices on box meshes have 49 / 14).  Every configuration is checked against the first."""
import os, statistics, sys
import ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oasisx_amd import fem, _lib, mesh as M
from oasisx_amd.la import SellMatrix
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
which = sys.argv[2] if len(sys.argv) > 2 else "p"
mesh = M.create_box(None, [[-1.,-1.,-1.],[1.,1.,1.]], [N,N,N])
deg, nc = {"p": (1, 1), "u": (2, 3), "u1": (2, 1)}[which]
V = fem.FunctionSpace(mesh, deg, window=int(os.environ.get("WINDOW", "4096")))
A = SellMatrix(V.pattern); A.vals.uniform_(0.5, 1.5)
npal = int(os.environ.get("PALETTE", "0"))
if npal:
    pal = torch.rand(npal, device="cuda", dtype=torch.float64) + 0.5
    A.vals.copy_(pal[torch.randint(0, npal, (A.vals.numel(),), device="cuda")])
    print("value dictionary built:", A.freeze(), "entries", A._struct.n_dict)
P = V.pattern
x = (torch.sin(torch.arange(P.n_cols*nc, device="cuda", dtype=torch.float64)*1e-3)+1).reshape(P.n_cols, nc).contiguous()
y = torch.zeros_like(x)
lib = _lib.load()
B = 12*P.nnz + 4*(P.n_rows+1) + nc*8*(P.n_cols+P.n_rows)
stored_old = P.size*((1 if A.vcode is not None else 8) + 2*P.frac16 + 4*(1-P.frac16)) + 8*(P.size//128) + nc*8*(P.n_cols+P.n_rows)
stored_pk = P.pk_groups*512*((1 if A.vcode is not None else 0) + 2) + (0 if A.vcode is not None else 8*P.size) + 64*P.pk_groups + nc*8*(P.n_cols+P.n_rows)
print(f"N={N} {which}: rows {P.n_rows} nnz {P.nnz} slots {P.size} groups {P.pk_groups} fallback slices {P.pk_fallback_slices}; "
      f"CSR bytes {B/1e6:.1f} MB, stored (r01 layout) {stored_old/1e6:.1f} MB, stored (packed) {stored_pk/1e6:.1f} MB; 16-bit coverage {P.frac16:.4f}")
configs = os.environ.get("CONFIGS", "v7;1,4,1,0;1,16,-1,0;1,8,-2,0;1,8,-4,0;9,16,-1,0;2,16,-1,17000" if npal else "v3;1,4,1,0;1,16,-1,0;1,8,-2,0;1,8,-4,0").split(";")
def apply(cfg):
    if cfg.startswith("v"):
        lib.ox_set_pk_mode(0, 0, 0, -1); lib.ox_set_spmv_variant(int(cfg[1:]))
    else:
        m, w, s, wm = (int(t) for t in cfg.split(","))
        lib.ox_set_pk_mode(m, w, s, wm)
res = {c: [] for c in configs}
ref = None
for c in configs:
    apply(c); y.zero_(); A.mult(x, y, nc); torch.cuda.synchronize()
    if not c.startswith("v"):
        nb, nf, wm = C.c_int(0), C.c_int(0), C.c_int(0)
        lib.ox_pk_plan_info(A.ref(), C.byref(nb), C.byref(nf), C.byref(wm))
        print(f"  config {c}: {nb.value} blocks, {nf.value} windows fit, window capacity {wm.value}")
    if ref is None: ref = y.clone()
    elif not c.startswith("9"):
        print(f"  config {c}: bit-identical to {configs[0]}: {bool(torch.equal(y, ref))}  (max |diff| {float((y-ref).abs().max()):.3e})")
for rnd in range(int(os.environ.get('ROUNDS', '7'))):
    for c in configs:
        apply(c)
        for _ in range(5): A.mult(x, y, nc)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = int(os.environ.get('REPS', '100'))
        e0.record()
        for _ in range(reps): A.mult(x, y, nc)
        e1.record(); torch.cuda.synchronize()
        res[c].append(e0.elapsed_time(e1)*1e3/reps)
for c in configs:
    med, mn = statistics.median(res[c]), min(res[c])
    sb = stored_old if c.startswith("v") else stored_pk
    print(f"config={c:16s} {which} N={N} median={med:.1f} us min={mn:.1f} us | CSR-priced {B/med/1e3:.0f} GB/s | stored bytes {sb/med/1e3:.0f} GB/s = {sb/med/1e3/8000:.3f} of 8 TB/s")
