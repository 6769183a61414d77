"""Per-rank set-up time and memory of a mesh-partitioned run, measured on ONE GPU: rank `r` of `P`
builds its partition metadata and its two spaces exactly as it would in the P-rank job (no
communication is involved in this part).   python tools/part_setup_time.py [N] [P] [r]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oasisx_amd import fem, mesh as M
from oasisx_amd.parallel import MeshPartition
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
r = int(sys.argv[3]) if len(sys.argv) > 3 else 0
def peak():
    """torch's own peak of the stage (the library's hipMalloc'd arrays are not in it: see `device in use`)"""
    torch.cuda.synchronize()
    g = torch.cuda.max_memory_allocated() / 2**30
    torch.cuda.reset_peak_memory_stats()
    return g
def in_use():
    torch.cuda.synchronize()
    torch.cuda.empty_cache()  # (cached blocks of torch's allocator are not "in use")
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**30
u0 = in_use()
torch.cuda.synchronize(); t0 = time.perf_counter()
mesh = M.create_box(None, [[-1., -1., -1.], [1., 1., 1.]], [N, N, N])
torch.cuda.synchronize(); t1 = time.perf_counter(); p1 = peak()
part = MeshPartition(mesh, r, P)
torch.cuda.synchronize(); t2 = time.perf_counter(); p2 = peak()
V = fem.FunctionSpace(mesh, 2, part=part)
torch.cuda.synchronize(); t3 = time.perf_counter(); p3 = peak()
Q = fem.FunctionSpace(mesh, 1, part=part)
torch.cuda.synchronize(); t4 = time.perf_counter(); p4 = peak()
print(f"stage peaks of torch's allocator (GiB): mesh {p1:.1f}, partition metadata {p2:.1f}, P2 space {p3:.1f}, P1 space {p4:.1f}; "
      f"device memory in use after set-up (hipMemGetInfo, library arrays included): {in_use() - u0:.1f} GiB")
print(f"N={N} rank {r} of {P}: mesh {t1-t0:.2f} s, partition metadata {t2-t1:.2f} s, P2 space {t3-t2:.2f} s "
      f"(owned {V.n_owned}, local {V.n_local}, cells {V.local_cells.numel()}), P1 space {t4-t3:.2f} s; "
      f"peak memory {max(p1, p2, p3, p4):.1f} GiB")
