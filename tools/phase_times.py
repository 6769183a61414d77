"""Wall-clock per phase of FractionalStep_AB_CN.solve (device-synchronised) -- diagnostics."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oasisx_amd as ox
from oasisx_amd import mesh as M
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nu = 0.01; dt = 0.005*32/N; clock = {"t": 0.0}
import math
xp = lambda x: torch if torch.is_tensor(x) else np   # as bench.py: device-side Dirichlet values
tg_u = lambda x,t: -xp(x).cos(np.pi*x[0])*xp(x).sin(np.pi*x[1])*math.exp(-2*nu*np.pi**2*t)
tg_v = lambda x,t: xp(x).cos(np.pi*x[1])*xp(x).sin(np.pi*x[0])*math.exp(-2*nu*np.pi**2*t)
tg_w = lambda x,t: xp(x).zeros_like(x[0])
tg_p = lambda x,t: -0.25*(np.cos(2*np.pi*x[0])+np.cos(2*np.pi*x[1]))*np.exp(-4*nu*np.pi**2*t)
onb = lambda x: np.isclose(np.abs(x[0]),1)|np.isclose(np.abs(x[1]),1)|np.isclose(np.abs(x[2]),1)
mesh = M.create_box(None, [[-1.,-1.,-1.],[1.,1.,1.]], [N,N,N])
fns=[tg_u,tg_v,tg_w]
def bcv(f):
    g = lambda x: f(x, clock["t"])
    g.supports_torch = True
    return g
bcs=[[ox.DirichletBC(bcv(f), ox.LocatorMethod.GEOMETRICAL, onb)] for f in fns]
ksp={"pc_type":"jacobi","ksp_rtol":1e-8,"ksp_atol":1e-14,"ksp_max_it":10000,"ksp_initial_guess_nonzero":True}
S=ox.FractionalStep_AB_CN(mesh,("Lagrange",2),("Lagrange",1),bcs_u=bcs,bcs_p=[],solver_options={"tentative":dict(ksp,ksp_type="bcgs"),"pressure":dict(ksp,ksp_type="cg"),"scalar":dict(ksp,ksp_type="cg")},
                          options={"low_memory_version": os.environ.get("MATRIX_FREE", "0") == "1"})
for i,f in enumerate(fns):
    S._u2[i].interpolate(lambda x,f=f:f(x,-dt)); S._u1[i].interpolate(lambda x,f=f:f(x,0.0))
S._p.interpolate(lambda x: tg_p(x,-dt/2))
def T():
    torch.cuda.synchronize(); return time.perf_counter()
acc = {}
def timed(name, fn):
    t0=T(); r=fn(); acc[name]=acc.get(name,0)+T()-t0; return r
for step in range(4):
    if step==1: acc.clear()
    clock["t"]+=dt
    timed("bc_update", lambda: [[bc.update_bc() for bc in b] for b in S._bcs_u])
    timed("assemble_first", lambda: S.assemble_first(dt,nu))
    timed("tent_assemble", S.velocity_tentative_assemble)
    timed("tent_solve", S.velocity_tentative_solve)
    timed("p_assemble", lambda: S.pressure_assemble(dt))
    timed("p_solve", lambda: S.pressure_solve(nu=nu))
    timed("update", lambda: S.velocity_update(dt))
    import oasisx_amd._lib as L
    lib=L.load(); st=L.current_stream(); n=S._n_u*3
    def shift():
        L.check(lib.ox_axpby(n,1.0,S._U1.ptr(),0.0,None,S._U2.ptr(),st)); L.check(lib.ox_axpby(n,1.0,S._U.ptr(),0.0,None,S._U1.ptr(),st)); L.check(lib.ox_axpby(S._n_q,1.0,S._PS.ptr(),0.0,None,S._P.ptr(),st))
    timed("shift", shift)
print({k: round(1e3*v/3,2) for k,v in acc.items()}, "sum", round(1e3*sum(acc.values())/3,1), S.iteration_counts())
