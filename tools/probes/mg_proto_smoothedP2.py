exec(open('tools/probes/mg_proto_smoothedP.py').read().split("run('mean (libplfx)'")[0])
def hierarchy3(smoothP, steps=1, galerkin_levels=0):
    levels = []
    nx, D6 = n, D0
    K = masked(Kf, dirichlet_mask(nx))
    while True:
        m = dirichlet_mask(nx)
        levels.append({'nx': nx, 'K': K, 'dinv': 1. / K.diagonal(), 'm': m})
        if nx % 2 or nx <= 2: break
        nc = nx // 2
        P = sp.diags(m) @ prolong(nc) @ sp.diags(dirichlet_mask(nc))
        for _ in range(steps):
            P = P - smoothP*sp.diags(levels[-1]['dinv']) @ (K @ P)
            P = sp.diags(m) @ P
        levels[-1]['P'] = P.tocsr()
        D6 = D6.reshape(nc, 2, nc, 2, 6).mean(axis=(1, 3)).reshape(-1, 6)
        if len(levels) <= galerkin_levels:
            Kc = (P.T @ K @ P).tocsr(); mc = dirichlet_mask(nc); Kc = (Kc + sp.diags(1. - mc)).tocsr()
        else:
            Kc = masked(assemble(nc, D6), dirichlet_mask(nc))
        K, nx = Kc, nc
    levels[-1]['lu'] = spla.splu(levels[-1]['K'].tocsc())
    return levels
for w in (0.3,0.5):
    try: run('mean coarse + smoothed transfers w=%.1f'%w, hierarchy3(w))
    except Exception as e: print('fail',e)
for gl in (1,2):
    run('galerkin on first %d levels, smoothedP .4, mean below'%gl, hierarchy3(0.4,1,gl))
run('galerkin all + 2-step smoothed P w=0.3', hierarchy3(0.3,2,99))
run('galerkin all + 2-step smoothed P w=0.45', hierarchy3(0.45,2,99))
