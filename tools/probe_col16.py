"""Probe (CPU): how many pairs of SELL-64 storage columns fit a single 16-bit column base --
the measurement behind the dual-base column codes.
Usage: PYTHONPATH=. python tools/probe_col16.py N degree"""
import torch, numpy as np, time, sys
from oasisx_amd import mesh as M, fem
N=int(sys.argv[1]); deg=int(sys.argv[2])
t=time.time()
m=M.create_box(None,[[-1,-1,-1],[1,1,1]],[N,N,N],device="cpu")
V=fem.FunctionSpace(m,deg)
P=V.pattern
print("build",time.time()-t,"n",P.n_rows,"nnz",P.nnz)
sp=P.slice_ptr.cpu().numpy(); cols=P.cols.cpu().numpy().astype(np.int64)
npair=len(cols)//128
c=cols.reshape(npair,128)
span=c.max(1)-c.min(1)
print("pairs",npair,"frac<65536",(span<65536).mean(),"frac<256",(span<256).mean(),"frac<4096",(span<4096).mean(), "max",span.max())
# per-slice
ps=(sp//128)
bad=np.zeros(len(sp)-1,bool)
badpair=np.nonzero(span>=65536)[0]
sl=np.searchsorted(ps,badpair,side='right')-1
bad[sl]=True
print("slices",len(bad),"incompressible",bad.mean())
# weight by entries
w=np.diff(sp)
print("entry fraction in incompressible slices",(w*bad).sum()/w.sum())
# alt: delta vs row
