"""Prototype: smoothed prolongation TRUNCATED to the bilinear (3x3-node) support, translation-preserving, Galerkin coarse
operators (stay 3x3 block stencils).  Usage: python tools/probes/mg_proto_truncP.py gpurun_out/tang128.npz"""
exec(open('tools/probes/mg_proto_smoothedP.py').read().split("run('mean (libplfx)'")[0])

def trunc_P(P1, P0, nx, nc):
    # pattern: node-level pattern of P0, all 2x2 dof combinations
    nf, ncn = (nx + 1) ** 2, (nc + 1) ** 2
    Pn = sp.csr_matrix((np.ones(P0.nnz), P0.indices, P0.indptr), shape=P0.shape)
    # node pattern: rows/cols are dof = 2*node + comp
    coo = P0.tocoo()
    rn, cn = coo.row // 2, coo.col // 2
    pat = sp.csr_matrix((np.ones(len(rn)), (rn, cn)), shape=(nf, ncn)); pat.data[:] = 1.
    W = sp.csr_matrix((coo.data, (rn, cn)), shape=(nf, ncn))      # summed twice (two comps) -> /2
    W.sum_duplicates(); W = W * 0.5
    patd = sp.kron(pat, np.ones((2, 2))).tocsr()
    Pt = P1.multiply(patd).tocsr()
    # translation preservation: sum over coarse nodes of the 2x2 blocks = I for every fine node
    E = sp.kron(sp.csr_matrix(np.ones((ncn, 1))), sp.identity(2)).tocsr()   # (2 ncn) x 2: translations
    S = (Pt @ E).toarray()                                        # (2 nf) x 2 block row sums
    defect = np.tile(np.eye(2), (nf, 1)) - S                      # what is missing
    # distribute by the P0 weights: B_ij += w_ij * defect_i
    Wc = W.tocoo()
    rows, cols, vals = [], [], []
    for a in range(2):
        for bcomp in range(2):
            rows.append(2 * Wc.row + a); cols.append(2 * Wc.col + bcomp); vals.append(Wc.data * defect[2 * Wc.row + a, bcomp])
    C = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=P1.shape)
    return (Pt + C).tocsr()

def hierarchy4(w, steps=1, trunc=True, fix_every=True):
    levels = []
    nx = n
    K = masked(Kf, dirichlet_mask(nx))
    while True:
        m = dirichlet_mask(nx)
        levels.append({'nx': nx, 'K': K, 'dinv': 1. / K.diagonal(), 'm': m})
        if nx % 2 or nx <= 2: break
        nc = nx // 2
        P0 = prolong(nc).tocsr()
        P = P0
        for _ in range(steps):
            P = P - w * sp.diags(levels[-1]['dinv'] * m) @ (K @ P)     # Dirichlet rows not smoothed
            if trunc: P = trunc_P(P.tocsr(), P0, nx, nc)
        P = sp.diags(m) @ P @ sp.diags(dirichlet_mask(nc))
        levels[-1]['P'] = P.tocsr()
        Kc = (P.T @ K @ P).tocsr(); mc = dirichlet_mask(nc); Kc = (Kc + sp.diags(1. - mc)).tocsr()
        K, nx = Kc, nc
    levels[-1]['lu'] = spla.splu(levels[-1]['K'].tocsc())
    print('   nnz per row of level-1 operator: %.1f' % (levels[1]['K'].nnz / levels[1]['K'].shape[0]))
    return levels

run('mean (libplfx)', hierarchy2('mean'))
for w in (0.3, 0.5, 0.65, 0.8):
    run('galerkin + truncated smoothed P w=%.2f' % w, hierarchy4(w))
for w in (0.3, 0.5, 0.65):
    run('galerkin + truncated smoothed P, 2 steps w=%.2f' % w, hierarchy4(w, 2))
    run('galerkin + truncated smoothed P, 4 steps w=%.2f' % w, hierarchy4(w, 4))
