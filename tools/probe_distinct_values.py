"""Probe (GPU): distinct bit patterns among the stored values of M, K and Ap on the box meshes --
the measurement behind the 1-byte value codes: 49 / 89 / 14, independent of N."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from tests.helpers import make_hip_problem
for N in (16, 32, 64):
    S, clock, mesh = make_hip_problem(3, N, u_deg=2, window=4096)
    for name, A in (("M", S._M), ("K", S._K), ("Ap", S._Ap)):
        v = A.vals
        u = torch.unique(v)
        # per-slice distinct max
        print(N, name, "slots", v.numel(), "distinct", u.numel(), flush=True)
    del S
