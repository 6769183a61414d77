"""assemble_first timing at the bench size (HIP events around the call, 10 repetitions)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oasisx_amd as ox
from oasisx_amd import mesh as M
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
mesh = M.create_box(None, [[-1.,-1.,-1.],[1.,1.,1.]], [N,N,N])
bcs = [[ox.DirichletBC(0.0, ox.LocatorMethod.GEOMETRICAL, lambda x: np.isclose(np.abs(x[0]), 1.0))] for _ in range(3)]
S = ox.FractionalStep_AB_CN(mesh, ("Lagrange", 2), ("Lagrange", 1), bcs_u=bcs, bcs_p=[], options={"low_memory_version": True})
S._U1.dev().normal_(); S._U2.dev().normal_()
for _ in range(3): S.assemble_first(0.01, 0.01)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): S.assemble_first(0.01, 0.01)
e1.record(); torch.cuda.synchronize()
print(f"OX_ASSEMBLE_U={os.environ.get('OX_ASSEMBLE_U','default')}: assemble_first {e0.elapsed_time(e1)/10:.3f} ms; checksum {float(S._A.vals.sum()):.12e} {float(S._BFIRST.dev().sum()):.12e}")
