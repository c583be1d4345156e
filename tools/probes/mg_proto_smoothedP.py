import sys, time
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
sys.argv = ['x', sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/tang128.npz']
src=open('tools/mg_proto.py').read()
src=src[:src.index("m0 = dirichlet_mask(n)")]
exec(src)
m0 = dirichlet_mask(n)
top = np.zeros((n + 1, n + 1, 2)); top[:, n, 1] = 1.
b = -(Kf @ top.ravel()) * m0
rng = np.random.default_rng(0)
b2 = rng.standard_normal(len(b)) * m0

def hierarchy2(kind, smoothP=0.0, nlev=99):
    levels = []
    nx, D6 = n, D0
    K = masked(Kf, dirichlet_mask(nx))
    while True:
        m = dirichlet_mask(nx)
        levels.append({'nx': nx, 'K': K, 'dinv': 1. / K.diagonal(), 'm': m})
        if nx % 2 or nx <= 2 or len(levels)>=nlev:
            break
        nc = nx // 2
        P = sp.diags(m) @ prolong(nc) @ sp.diags(dirichlet_mask(nc))
        if smoothP>0:
            P = P - smoothP*sp.diags(levels[-1]['dinv']) @ (K @ P)
            P = sp.diags(m) @ P
        levels[-1]['P'] = P.tocsr()
        if kind == 'galerkin':
            Kc = (P.T @ K @ P).tocsr()
            mc = dirichlet_mask(nc)
            Kc = (Kc + sp.diags(1. - mc)).tocsr()
        else:
            D6 = D6.reshape(nc, 2, nc, 2, 6).mean(axis=(1, 3)).reshape(-1, 6)
            Kc = masked(assemble(nc, D6), dirichlet_mask(nc))
        K, nx = Kc, nc
    levels[-1]['lu'] = spla.splu(levels[-1]['K'].tocsc())
    return levels

def run(tag, lv, vc=None):
    global vcycle
    old=vcycle
    if vc: vcycle=vc
    t=time.time()
    print('%-40s its tension %3d random %3d  (%.1fs)'%(tag, pcg(lv,b), pcg(lv,b2), time.time()-t), flush=True)
    vcycle=old

run('mean (libplfx)', hierarchy2('mean'))
run('galerkin', hierarchy2('galerkin'))
for w in (0.3,0.5,0.65):
    run('galerkin + smoothed P w=%.2f'%w, hierarchy2('galerkin', w))
# Chebyshev smoother degree k on D^-1 K with lmax ~ 2.3
def cheb_vcycle(deg, lmax=2.4, lmin_frac=0.25):
    def smooth(L, x, bb):
        lmin=lmax*lmin_frac
        theta=(lmax+lmin)/2; delta=(lmax-lmin)/2; sigma=theta/delta; rho=1/sigma
        r = L['dinv']*(bb - L['K']@x)
        d = r/theta
        x = x + d
        for k in range(1,deg):
            rho_new = 1/(2*sigma-rho)
            r = L['dinv']*(bb - L['K']@x)
            d = rho_new*rho*d + 2*rho_new/delta*r
            x = x + d
            rho=rho_new
        return x
    def vc(levels,l,bb,om=0.65,nu=2):
        L=levels[l]
        if 'lu' in L: return L['lu'].solve(bb)
        x=smooth(L,np.zeros_like(bb),bb)
        r=bb-L['K']@x
        x+=L['P']@vc(levels,l+1,L['P'].T@r)
        return smooth(L,x,bb)
    return vc
for deg in (2,4):
    run('mean + chebyshev deg %d'%deg, hierarchy2('mean'), cheb_vcycle(deg))
    run('galerkin smoothedP 0.5 + cheb deg %d'%deg, hierarchy2('galerkin',0.5), cheb_vcycle(deg))
