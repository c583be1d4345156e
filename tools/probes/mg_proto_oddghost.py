"""scipy prototype (CPU): multigrid on a Q4 elasticity grid with an ODD number of elements in y and a Dirichlet edge on that side.
Compares, as preconditioners of CG (V(2,2), damped Jacobi 0.65, coarse operators re-discretised from mean generators as in libplfx):
  even     the neighbouring even mesh (reference iteration count)
  ghost    ceil-halving with a zero-stiffness ghost element beyond the edge (what round 5 built first): sum of children / 4,
           ghost line + last coincident line masked like the edge, transfers unchanged
  half     ceil-halving where the last coarse element has HALF height and ends at the edge: mean over the existing children,
           last coarse node line coincident with the fine edge line (prolongation weight 1 there), element geometry 2h x h
python tools/probes/mg_proto_oddghost.py"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

E, NU = 200e3, 0.3
C = E / ((1 + NU) * (1 - 2 * NU)) * np.array([[1 - NU, NU, 0], [NU, 1 - NU, 0], [0, 0, (1 - 2 * NU) / 2]])


def elem_K(lx, ly, scale=1.):
    """8x8 plane-strain Q4 stiffness, node order (j,k),(j,k+1),(j+1,k),(j+1,k+1)"""
    g = 1 / np.sqrt(3)
    xs = np.array([-1., -1., 1., 1.]); ys = np.array([-1., 1., -1., 1.])
    K = np.zeros((8, 8))
    for xi in (-g, g):
        for eta in (-g, g):
            dNx = xs * (1 + ys * eta) / 4 * 2 / lx
            dNy = ys * (1 + xs * xi) / 4 * 2 / ly
            B = np.zeros((3, 8))
            B[0, 0::2] = dNx; B[1, 1::2] = dNy; B[2, 0::2] = dNy; B[2, 1::2] = dNx
            K += B.T @ C @ B * (lx * ly / 4)
    return scale * K


def assemble(nx, ny, lxs, lys, scale):
    """lxs[j], lys[k]: element sizes; scale[j,k]: stiffness factor"""
    nd = 2 * (nx + 1) * (ny + 1)
    rows, cols, vals = [], [], []
    cache = {}
    for j in range(nx):
        for k in range(ny):
            key = (lxs[j], lys[k])
            if key not in cache:
                cache[key] = elem_K(*key)
            n1 = j * (ny + 1) + k
            nodes = [n1, n1 + 1, n1 + ny + 1, n1 + ny + 2]
            dofs = np.array([[2 * n, 2 * n + 1] for n in nodes]).ravel()
            rows.append(np.repeat(dofs, 8)); cols.append(np.tile(dofs, 8)); vals.append((scale[j, k] * cache[key]).ravel())
    return sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(nd, nd)).tocsr()


def mask_fine(nx, ny):
    m = np.ones((nx + 1, ny + 1, 2))
    m[0, :, 0] = 0; m[:, 0, 1] = 0; m[:, ny, 1] = 0     # left ux, bottom uy, top uy (prescribed)
    return m


def prolong(nxc, nyc, nxf, nyf, half_x, half_y):
    """bilinear P (fine nodes x coarse nodes, per component).  half_*: the last coarse line coincides with the fine edge line"""
    def p1(nc, nf, half):
        P = np.zeros((nf + 1, nc + 1))
        for i in range(nf + 1):
            if half and i == nf:
                P[i, nc] = 1.
            elif i % 2 == 0:
                P[i, i // 2] = 1.
            else:
                P[i, i // 2] = 0.5
                if i // 2 + 1 <= nc:
                    P[i, i // 2 + 1] = 0.5
        return sp.csr_matrix(P)
    Px, Py = p1(nxc, nxf, half_x), p1(nyc, nyf, half_y)
    return sp.kron(sp.kron(Px, Py), sp.identity(2)).tocsr()


def build(nx, ny, mode):
    levels = []
    lxs, lys = np.ones(nx), np.ones(ny)
    scale = np.ones((nx, ny))
    m = mask_fine(nx, ny)
    while True:
        K = assemble(nx, ny, lxs, lys, scale)
        levels.append(dict(K=K, m=m.ravel().copy(), nx=nx, ny=ny))
        if nx * ny <= 4:
            break
        nxc, nyc = (nx + 1) // 2, (ny + 1) // 2
        ox, oy = nx % 2 == 1, ny % 2 == 1
        # coarse element sizes and generators (stiffness scale): mean of children
        sc = np.zeros((nxc, nyc)); cnt = np.zeros((nxc, nyc))
        for j in range(nx):
            for k in range(ny):
                sc[j // 2, k // 2] += scale[j, k]; cnt[j // 2, k // 2] += 1
        if mode == 'half':
            sc = sc / cnt
            lxc = np.array([lxs[2 * J] + (lxs[2 * J + 1] if 2 * J + 1 < nx else 0.) for J in range(nxc)])
            lyc = np.array([lys[2 * K] + (lys[2 * K + 1] if 2 * K + 1 < ny else 0.) for K in range(nyc)])
        else:
            sc = sc / 4.
            lxc, lyc = np.full(nxc, 2 * lxs[0]), np.full(nyc, 2 * lys[0])
        half = mode == 'half'
        P = prolong(nxc, nyc, nx, ny, half and ox, half and oy)
        # coarse mask
        mc = np.ones((nxc + 1, nyc + 1, 2))
        for J in range(nxc + 1):
            for Kk in range(nyc + 1):
                jf, kf = min(2 * J, nx), min(2 * Kk, ny)
                mm = m[jf, kf].copy()
                if mode == 'ghost':
                    if 2 * J + 1 == nx: mm = np.minimum(mm, m[nx, kf])
                    if 2 * Kk + 1 == ny: mm = np.minimum(mm, m[jf, ny])
                mc[J, Kk] = mm
        levels[-1]['P'] = P
        nx, ny, lxs, lys, scale, m = nxc, nyc, lxc, lyc, sc, mc
    return levels


def vcycle(levels, l, b, nu=2, om=0.65):
    L = levels[l]
    K, m = L['K'], L['m']
    if l == len(levels) - 1:
        free = np.where(m > 0)[0]
        x = np.zeros_like(b)
        x[free] = np.linalg.solve(K[free][:, free].toarray(), b[free])
        return x
    dinv = m / K.diagonal()
    x = np.zeros_like(b)
    for _ in range(nu):
        x = x + om * dinv * (b - K @ x)
    r = m * (b - K @ x)
    P = L['P']
    bc = levels[l + 1]['m'] * (P.T @ r)
    x = x + m * (P @ vcycle(levels, l + 1, bc))
    for _ in range(nu):
        x = x + om * dinv * (b - K @ x)
    return x


def pcg_its(levels, rtol=1e-10, seed=0):
    L = levels[0]
    K, m = L['K'], L['m']
    rng = np.random.default_rng(seed)
    b = m * rng.standard_normal(K.shape[0])
    x = np.zeros_like(b); r = b.copy(); z = vcycle(levels, 0, r); p = z.copy(); rz = r @ z
    for it in range(1, 2000):
        q = m * (K @ p)
        a = rz / (p @ q)
        x += a * p; r -= a * q
        if np.linalg.norm(r) <= rtol * np.linalg.norm(b):
            return it
        z = vcycle(levels, 0, r); rzn = r @ z
        p = z + rzn / rz * p; rz = rzn
    return -1


if __name__ == '__main__':
    for (nx, ny) in ((32, 32), (32, 31), (31, 32), (31, 31), (64, 63), (48, 47)):
        out = []
        for mode in ('ghost', 'half'):
            lv = build(nx, ny, mode)
            out.append('%s %3d' % (mode, pcg_its(lv)))
        print('%3d x %3d (%d levels): ' % (nx, ny, len(lv)) + '   '.join(out))
