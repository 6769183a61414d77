"""How regular are the gathers of the SELL-64 SpMV?  For the P2 (or P1) pattern of an N^3 box mesh: the share of
(slice, entry k) positions whose 64 lanes read 64 CONSECUTIVE columns (one coalesced run), and the run lengths of
entries that continue the previous entry's columns by +1 in every lane (candidates for sharing one load)."""
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
tot = cons = shift = 0
hist = {}
for s in range(0, P.n_slices, max(1, P.n_slices // 400)):  # a sample of slices
    b, e = int(sp[s]), int(sp[s + 1])
    w = (e - b) // 64
    c = cols[b:e].reshape(w // 2, 64, 2).permute(0, 2, 1).reshape(w, 64).to(torch.int64)  # [k][lane]
    rl = P.row_len[s * 64:(s + 1) * 64].to(torch.int64)
    if rl.numel() < 64:
        continue
    lane = torch.arange(64, device=c.device)
    run = 0
    for k in range(int(rl.min())):
        tot += 1
        ck = c[k]
        is_cons = bool((ck - ck[0] == lane).all())
        cons += is_cons
        cont = k > 0 and bool((ck == c[k - 1] + 1).all())
        shift += cont
        if cont:
            run += 1
        else:
            if k > 0:
                hist[run + 1] = hist.get(run + 1, 0) + 1
            run = 0
    hist[run + 1] = hist.get(run + 1, 0) + 1
print(f"N={N} degree {deg}: {tot} (slice, entry) positions sampled; lanes read 64 consecutive columns in {cons / tot:.3f}; "
      f"entry continues the previous one by +1 in every lane in {shift / tot:.3f}")
print("run lengths (entries sharing one contiguous window):", dict(sorted(hist.items())))
n_runs = sum(hist.values())
print(f"loads if every run were ONE load: {n_runs / tot:.3f} of the gathers")
