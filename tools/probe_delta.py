"""Share of (slice, entry k, lane group) positions of the SELL-64 pattern whose columns are ROW + one common delta
(what a per-group delta descriptor could replace), for lane groups of 64 / 32 / 16 / 8, on the N^3 box mesh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oasisx_amd import fem, mesh as M
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 2
mesh = M.create_box(None, [[-1., -1., -1.], [1., 1., 1.]], [N, N, N])
V = fem.FunctionSpace(mesh, deg)
P = V.pattern
sp = P.slice_ptr.cpu()
cols = P.cols
res = {g: [0, 0] for g in (64, 32, 16, 8)}
nnz_tot = 0
for s in range(0, P.n_slices, max(1, P.n_slices // 600)):
    b, e = int(sp[s]), int(sp[s + 1])
    w = (e - b) // 64
    c = cols[b:e].reshape(w // 2, 64, 2).permute(0, 2, 1).reshape(w, 64).to(torch.int64)  # [k][lane]
    rl = P.row_len[s * 64:(s + 1) * 64].to(torch.int64)
    if rl.numel() < 64:
        continue
    lane = torch.arange(64)
    valid = torch.arange(w)[:, None] < rl[None, :]  # [k][lane]
    d = c - (s * 64 + lane)[None, :]
    nnz_tot += int(valid.sum())
    for g in res:
        dg = d.reshape(w, 64 // g, g)
        vg = valid.reshape(w, 64 // g, g)
        big = torch.where(vg, dg, torch.full_like(dg, -10**12)).amax(-1)
        sml = torch.where(vg, dg, torch.full_like(dg, 10**12)).amin(-1)
        anyv = vg.any(-1)
        uni = anyv & (big == sml)
        res[g][0] += int((vg & uni[..., None]).sum())  # nonzeros covered by a uniform group
        res[g][1] += int(anyv.sum())  # descriptors needed
print(f"N={N} degree {deg}: {nnz_tot} nonzeros sampled")
for g, (cov, nd) in res.items():
    print(f"  lane groups of {g}: {cov / nnz_tot:.3f} of the nonzeros sit in a uniform group ({nd} groups with any entry)")
