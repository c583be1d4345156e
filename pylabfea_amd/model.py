"""``Model`` façade: the reference's FE-model API with the hot path running in libplfx on the MI355X.

Mirrors pylabfea.Model (reference /root/reference/src/pylabfea/model.py) for 2-d models with linear
Q4 elements: ``geom`` (:514), ``assign`` (:553), ``bcleft/bcright/bcbot/bctop/bcnode`` (:577-757),
``mesh`` (:758), ``setupK`` (:954), ``solve`` (:979), ``bcval`` (:1452), ``calc_global`` (:1473) and
the per-element results ``element[i].sig/eps/epl``.  Index generation (node numbering, connectivity,
boundary sets, free-DOF list) is bit-identical to the reference; the load-step / K-iteration control
flow of ``solve`` is restated here in Python and every O(Nel) / O(Ndof) operation is one C-ABI call:

    assemble (setupK) -> apply_bc (calc_BC) -> solve (Kred + np.linalg.solve, here Jacobi-PCG on the
    block-ELL matrix) -> sweep (Material.response per element + tangent refresh) -> update_state.

Out of scope: 1-d models, quadratic elements (the reference raises NotImplementedError for 2-d
quadratic, model.py:361), user-supplied node positions, plotting.
"""
import os
import warnings

import numpy as np

from . import _lib


class _ElementView(object):
    """Read-only view of one element's results (``Model.element[i]``), backed by the arrays that
    ``Model`` downloads from HBM after ``solve``."""

    def __init__(self, model, i):
        self.Model = model
        self.index = i

    @property
    def nodes(self):
        return [int(n) for n in self.Model._conn[self.index]]

    @property
    def Mat(self):
        return self.Model.mat[self.Model._mat_id[self.index]]

    @property
    def Lelx(self):
        return float(self.Model._lxy[self.index, 0])

    @property
    def Lely(self):
        return float(self.Model._lxy[self.index, 1])

    @property
    def Vel(self):
        return self.Lelx * self.Lely * self.Model.thick

    @property
    def Jac(self):
        return 4. * self.Vel

    @property
    def CV(self):
        return self.Model._element_CV(self.Mat)

    @property
    def sig(self):
        return self.Model._state('sig')[self.index]

    @property
    def eps(self):
        return self.Model._state('eps')[self.index]

    @property
    def epl(self):
        return self.Model._state('epl')[self.index]

    @property
    def elstiff(self):
        return self.Model._state('elstiff')[self.index].reshape(6, 6)

    @property
    def res_sig(self):
        return self.Model._state('res_sig')[self.index]

    @property
    def res_depl(self):
        return self.Model._state('res_depl')[self.index]

    @property
    def Bmat(self):
        B = self.Model._engine.get_bmat(self.index)
        return [B[g] for g in range(4)]

    @property
    def Kel(self):
        return self.Model._engine.get_kel(self.index)

    @property
    def stat_nlin(self):
        return {'max_iter': 0, 'max_steps': int(self.Model._state('max_steps')[self.index]), 'max_dstiff': None}

    def node_num(self):
        ind = []
        for j in self.nodes:
            ind.extend([2 * j, 2 * j + 1])
        return ind

    def eps_t(self):
        return sum(B @ self.Model.u[self.node_num()] for B in self.Bmat)

    def deps(self):
        return sum(B @ self.Model.du[self.node_num()] for B in self.Bmat)

    def dsig(self):
        return self.elstiff @ self.deps()


class _ElementList(object):
    def __init__(self, model, n):
        self._m = model
        self._n = n

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [_ElementView(self._m, k) for k in range(*i.indices(self._n))]
        i = int(i)
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError('element index out of range')
        return _ElementView(self._m, i)

    def __iter__(self):
        for k in range(self._n):
            yield _ElementView(self._m, k)


class Model(object):
    """Finite-element model (2-d, Q4) with the reference's attributes: ``dim, planestress, Nsec, LS,
    lenx, leny, thick, nonlin, mat, Nnode, NnodeX, NnodeY, Nel, Ndof, npos, noleft, noright, nobot,
    notop, noinner, element, u, f, du, sgl, egl, epgl, glob, nsteps, niter, co_nconv``.

    Extra knobs (not in the reference): ``device`` (GPU ordinal), ``cg_rtol`` / ``cg_maxit`` for the
    iterative solve that replaces the dense LU, ``solver_stats`` (PCG iterations of every solve).
    """

    def __init__(self, dim=1, planestress=False, device=0):
        if dim not in (1, 2):
            raise ValueError('dim must be either 1 or 2')
        if dim == 1:
            raise NotImplementedError('pylabfea_amd builds the 2-d Q4 path only (dim=2)')
        self.dim = dim
        self.planestress = planestress
        self.device = device
        self.bcl = np.zeros(dim)
        self.bcb = np.zeros(dim)
        self.bct = np.zeros(dim)
        self.bcr = np.zeros(dim)
        self.bcn = np.zeros(dim)
        self.noset = None
        self.ubctop = [False, False]
        self.ubcright = [False, False]
        self.ubcleft = [True, False]
        self.ubcbot = [False, True]
        self.ubcn = [False, False]
        self.nonlin = False
        self.sgl = np.zeros((1, 6))
        self.egl = np.zeros((1, 6))
        self.epgl = np.zeros((1, 6))
        self.u = None
        self.f = None
        self.du = None
        self.Nnode = None
        self.glob = {'ebc1': None, 'ebc2': None, 'sbc1': None, 'sbc2': None,
                     'eps': np.zeros(6), 'sig': np.zeros(6), 'epl': np.zeros(6)}
        self.cg_rtol = float(os.environ.get('PLFX_CG_RTOL', '1e-10'))  # relative residual |r|/|b| of the PCG solve;
        # measured field error vs the reference's dense LU ~ 4 x cg_rtol (4e-10 here, north-star budget 1e-6)
        self.cg_maxit = 100000
        self.precond = None          # None: library default (multigrid when available), 0 Jacobi, 1 multigrid
        self.operator = None         # None: library default (matrix-free on uniform structured grids), 0 assembled, 1 matrix-free
        self.solver_stats = []
        self.n_sweeps = 0            # material sweeps (K-iterations) executed so far
        self._engine = None
        self._cache = {}
        self._bnd_idx = None
        self._bnd_offs = None
        self._bc_struct = None
        self._bc_registered = None
        self._dev_coll = False
        self._native_step = os.environ.get('PLFX_NATIVE_STEP', '1') != '0'
        self._step_io = _lib.CStep()
        self._defer_finish = os.environ.get('PLFX_DEFER_FINISH', '1') != '0'
        self._shard = None  # (rank, nranks, uid)
        self._strip = None  # strip-local engine: column window and numbering offsets of this rank (strip_plan)
        self._max_load_steps = None  # benchmarking aid: stop after this many load steps
        self._step_hook = None       # benchmarking aid: called as hook(il) after every load step
        # work-hardening SVC materials: 'sequential' = the reference's semantics (ONE hardening modulus per Material object,
        # handed from element to element in index order, material.py:808-814 + model.py:1340-1359; one GPU), 'per_point' = one
        # modulus per material point (faster: no repeated sweeps; the only form on several GPUs; ~1e-4 off the reference)
        self.wh_carry = 'sequential'

    # ------------------------------------------------------------------ pre-processing
    def geom(self, sect=1, LX=None, LY=1., LZ=1.):
        """Geometry and sections (model.py:514-551)."""
        if type(sect) == list:
            self.Nsec = len(sect)
            self.LS = np.array(sect)
            self.lenx = sum(sect)
        elif type(sect) == int:
            if sect < 1:
                raise ValueError('At least one section must be defined.')
            if LX is None:
                raise ValueError('LX must be given if sect is of type int')
            self.lenx = LX
            self.Nsec = sect
            self.LS = np.ones(sect) * self.lenx / sect
        else:
            raise TypeError('Sect must be either list or int, not {}'.format(type(sect)))
        self.leny = LY
        self.thick = LZ

    def assign(self, mats):
        """Assign a material to each section (model.py:553-575)."""
        if len(mats) != self.Nsec:
            raise ValueError('Numer of materials ({}) does not match number of sections ({})'
                             .format(len(mats), self.Nsec))
        self.mat = mats
        self.nonlin = False
        for mat in mats:
            if mat.sy is not None:
                self.nonlin = True

    @staticmethod
    def _dir(bcdir, who):
        if (isinstance(bcdir, str) and bcdir.lower() == 'x') or bcdir == 0:
            return 0
        if (isinstance(bcdir, str) and bcdir.lower() == 'y') or bcdir == 1:
            return 1
        raise ValueError('{}: Unknown value for direction: {}'.format(who, bcdir))

    def bcleft(self, val=0., bctype='disp', bcdir='x'):
        """BC on lhs nodes (model.py:577-611)."""
        j = self._dir(bcdir, 'bcleft')
        self.bcl[j] = val
        if bctype.lower() == 'disp':
            self.ubcleft[j] = True
        elif bctype.lower() == 'force':
            self.ubcleft[j] = False
            if np.abs(val) > 1.e-6:
                raise ValueError('Finite force values at left boundary not supported.')
        else:
            raise ValueError('bcleft: Unknown BC: %s' % bctype)

    def bcright(self, val, bctype, bcdir='x'):
        """BC on rhs nodes (model.py:613-645)."""
        j = self._dir(bcdir, 'bcright')
        self.bcr[j] = val
        if bctype.lower() == 'disp':
            self.ubcright[j] = True
        elif bctype.lower() == 'force':
            self.ubcright[j] = False
        else:
            raise TypeError('bcright: Unknown BC: {}'.format(bctype))

    def bcbot(self, val=0., bctype='disp', bcdir='y'):
        """BC on bottom nodes (model.py:647-683)."""
        j = self._dir(bcdir, 'bcbot')
        self.bcb[j] = val
        if bctype.lower() == 'disp':
            self.ubcbot[j] = True
        elif bctype.lower() == 'force':
            self.ubcbot[j] = False
            if np.abs(val) > 1.e-6:
                raise ValueError('Finite force values at bottom boundary not supported.')
        else:
            raise ValueError('bcbot: Unknown BC: {}'.format(bctype))

    def bctop(self, val, bctype, bcdir='y'):
        """BC on top nodes (model.py:685-717)."""
        j = self._dir(bcdir, 'bctop')
        self.bct[j] = val
        if bctype.lower() == 'disp':
            self.ubctop[j] = True
        elif bctype.lower() == 'force':
            self.ubctop[j] = False
        else:
            raise TypeError('bctop: Unknown BC: {}'.format(bctype))

    def bcnode(self, node, val, bctype, bcdir):
        """BC on a freely defined node set (model.py:719-757)."""
        # the reference keeps `node` as given (int, list or 1-element array); normalise to ints
        self.noset = [int(n) for n in np.ravel(node)]
        j = self._dir(bcdir, 'bcnode')
        self.bcn[j] = val
        if bctype.lower() == 'disp':
            self.ubcn[j] = True
        elif bctype.lower() == 'force':
            self.ubcn[j] = False
        else:
            raise TypeError('bcnode: Unknown BC: {}'.format(bctype))

    def mesh(self, elmts=None, nodes=None, NX=10, NY=1, SF=1):
        """Structured Q4 mesh (model.py:758-952): node ``j*NnodeY + k`` (x-index slow), element
        ``j*NY + k``, connectivity ``[n1, n1+1, n1+NnodeY, n1+NnodeY+1]``; laminate sections or an
        ``elmts`` material-id array.  All integer products equal the reference's bit for bit."""
        if SF != 1:
            raise NotImplementedError('Error: Quadrilateral elements with quadratic shape function not yet implemented')
        if nodes is not None:
            raise NotImplementedError('mesh: user-supplied node positions are not supported')
        self.shapefact = SF
        if elmts is not None:
            el = np.array(elmts, dtype=int)
            if len(el.shape) != self.dim:
                raise ValueError('Cannot use a {}-shaped mesh with a {}-dimemsional model'.format(el.shape, self.dim))
            NX, NY = el.shape
        if NX < self.Nsec:
            raise TypeError('Error: Number of elements is smaller than number of sections')
        if self.u is not None:
            warnings.warn('Warning: Solution of previous steps is deleted')
            self.u = None
            self.f = None
        self.NnodeX = NX + 1
        self.NnodeY = NY + 1
        self.Nnode = self.NnodeX * self.NnodeY
        self.Ndof = self.Nnode * 2
        self.Nel = NX * NY
        nrow = self.NnodeY
        # The grid is kept as a DESCRIPTION (round 4): node x-positions per column, element width / material per column (or the
        # `elmts` array), one element height.  The mesh-sized products of the reference -- `npos`, the connectivity, per-element
        # sizes and material numbers -- are materialised on first access (properties below, same numbers as before, bit for
        # bit), and the library writes its own index arrays from the description (plfx_set_mesh_structured): mesh() no longer
        # costs 155 ms of NumPy index work at 1024^2 before the first load step can start.
        dy = self.leny / NY
        if elmts is None:
            # elements per section: proportional, the largest section absorbs the remainder (:826-830)
            hh = self.LS / self.lenx
            nes = [int(x) for x in np.round(hh * NX)]
            if np.sum(nes) != NX:
                im = np.argmax(self.LS)
                nes[im] = nes[im] - np.sum(nes) + NX
            xcol = np.zeros(self.NnodeX)
            mat_col = np.zeros(NX, dtype=np.int64)
            dx_col = np.zeros(NX)
            c0 = 0
            for i in range(self.Nsec):
                dx = self.LS[i] / nes[i]
                j0 = 0 if i == 0 else 1
                jl = np.arange(j0, nes[i] + 1)
                # x-position uses the section's own dx times the global column index (:847)
                xcol[c0 + jl] = (jl + c0) * dx
                mat_col[c0:c0 + nes[i]] = i
                dx_col[c0:c0 + nes[i]] = dx
                c0 += nes[i]
            mat_el = None
        else:
            dx = self.lenx / NX
            xcol = np.arange(self.NnodeX) * dx
            dx_col = np.full(NX, dx)
            mat_col = None
            mat_el = (el - 1).ravel().astype(np.int64)
            if mat_el.min() < 0 or mat_el.max() >= len(self.mat):
                raise IndexError('mesh: material number in elmts out of range')
        self._grid = {'xcol': xcol, 'dx_col': dx_col, 'dy': dy, 'mat_col': mat_col, 'mat_el': mat_el}
        self._npos_arr = self._conn_arr = self._lxy_arr = self._mat_id_arr = None
        # boundary node lists in the reference's append order (:897-911): j outer, k inner
        self.noleft = list(range(nrow))
        self.noright = list(range(NX * nrow, NX * nrow + nrow))
        self.nobot = list(range(0, self.NnodeX * nrow, nrow))
        self.notop = list(range(NY, self.NnodeX * nrow, nrow))
        self._noinner = None          # (NX-1)(NY-1) entries: materialised on first access (property noinner)
        self._NX, self._NY = NX, NY
        self.element = _ElementList(self, self.Nel)
        self._bnd_idx = None
        self._bnd_offs = None
        self._bc_struct = None
        self._bc_registered = None
        self._drop_engine()

    # -- mesh-sized products of Model.mesh, materialised on first access
    _grid = None
    _npos_arr = _conn_arr = _lxy_arr = _mat_id_arr = None

    @property
    def npos(self):
        """nodal positions, interleaved (x, y) in node order j * NnodeY + k (model.py:893, :847)"""
        if self._npos_arr is None and getattr(self, '_grid', None) is not None:
            g, nrow = self._grid, self.NnodeY
            npos = np.empty(self.Ndof)
            npos[1::2] = np.tile(np.arange(nrow) * g['dy'], self.NnodeX)
            npos[0::2] = np.repeat(g['xcol'], nrow)
            self._npos_arr = npos
        return self._npos_arr

    @npos.setter
    def npos(self, v):
        self._npos_arr = v

    def _node_coord(self, nodes, pos):
        """coordinate `pos` (0 = x, 1 = y) of the given nodes without materialising `npos`"""
        g = self._grid
        nodes = np.asarray(nodes, dtype=np.int64)
        return g['xcol'][nodes // self.NnodeY] if pos == 0 else (nodes % self.NnodeY) * g['dy']

    @property
    def _conn(self):
        """connectivity [n1, n1+1, n1+nrow, n1+nrow+1], n1 = (ih // NY) * nrow + ih % NY (:936-948); int32 like the library"""
        if self._conn_arr is None and self._grid is not None:
            NY, nrow = self._NY, self.NnodeY
            ih = np.arange(self.Nel, dtype=np.int32)
            n1 = (ih // NY) * nrow + ih % NY
            conn = np.empty((self.Nel, 4), dtype=np.int32)
            conn[:, 0] = n1
            conn[:, 1] = n1 + 1
            conn[:, 2] = n1 + nrow
            conn[:, 3] = n1 + nrow + 1
            self._conn_arr = conn
        return self._conn_arr

    @property
    def _lxy(self):
        if self._lxy_arr is None and self._grid is not None:
            lxy = np.empty((self.Nel, 2))
            lxy[:, 0] = np.repeat(self._grid['dx_col'], self._NY)
            lxy[:, 1] = self._grid['dy']
            self._lxy_arr = lxy
        return self._lxy_arr

    @property
    def _mat_id(self):
        if self._mat_id_arr is None and self._grid is not None:
            g = self._grid
            self._mat_id_arr = g['mat_el'] if g['mat_el'] is not None else np.repeat(g['mat_col'], self._NY)
        return self._mat_id_arr

    @property
    def noinner(self):
        """nodes that are on no edge, in the reference's append order (model.py:897-911: j outer, k inner)"""
        if self._noinner is None:
            nrow = self.NnodeY
            j, k = np.meshgrid(np.arange(1, self.NnodeX - 1), np.arange(1, nrow - 1), indexing='ij')
            self._noinner = (j * nrow + k).ravel().tolist()
        return self._noinner

    @noinner.setter
    def noinner(self, v):
        self._noinner = v

    # ------------------------------------------------------------------ engine plumbing
    def _drop_engine(self):
        if self._engine is not None:
            self._engine.close()
        self._engine = None
        self._cache = {}

    def _element_CV(self, mat):
        """Element elastic matrix (Element.__init__, model.py:272-303)."""
        if self.planestress:
            hh = mat.E / (1 - mat.nu * mat.nu)
            C12 = mat.nu * hh
            C11 = hh
            return np.array([[C11, C12, 0., 0., 0., 0.],
                             [C12, C11, 0., 0., 0., 0.],
                             [0., 0., 0., 0., 0., 0.],
                             [0., 0., 0., 0., 0., 0.],
                             [0., 0., 0., 0., 0., 0.],
                             [0., 0., 0., 0., 0., mat.C44]])
        return np.array(mat.CV, dtype=float)

    def distribute(self, rank, nranks, uid, host_allreduce=None, mode=None, coarse_level=None):
        """Run this model on ``nranks`` GPUs, one process per GPU; this process is ``rank`` on GPU ``self.device``.
        ``uid`` is the RCCL unique id created on rank 0 (``_lib.Context.comm_unique_id``) and broadcast by the caller;
        ``uid=None`` with ``host_allreduce=fn(array, op)`` selects the host-staged transport instead (tests on one GPU;
        ``pylabfea_amd.host_transport`` builds ``fn`` from a ``torch.distributed`` process group).

        ``mode='strip'`` (default wherever it applies: uniform structured grid, strips at least one halo wide): every rank
        holds its x-strip of element columns plus a halo as a standalone local problem -- state, operator, multigrid
        levels and solve are all distributed, the ranks exchange one halo slab and a coarse right-hand side per PCG
        iteration (plfx_set_strip).  ``mode='replicated'``: the material state and sweep are sharded by x-strips, the
        operator and the solve are replicated (any mesh).  After ``solve`` the arrays ``u, f, du`` and the element
        results are filled on this rank's columns only (zeros elsewhere)."""
        if mode not in (None, 'strip', 'replicated'):
            raise ValueError("distribute: mode must be None, 'strip' or 'replicated'")
        if int(nranks) > 1 and uid is None and host_allreduce is None:
            raise ValueError('distribute: pass the RCCL unique id (uid) or a host_allreduce callback')
        self._shard = (int(rank), int(nranks), uid)
        self._host_allreduce = host_allreduce
        self._dist_mode = mode
        self._strip_level = coarse_level
        self._strip = None
        self._bc_struct = None
        self._bc_registered = None
        self._bnd_idx = None
        self._bnd_offs = None
        self._drop_engine()

    def strip_plan(self, rank, nranks, coarse_level=None):
        """Column ranges of the strip-local engine for ``rank`` of ``nranks``: dict with the owned element columns
        ``c0, c1``, the local window ``g0, g1`` (owned + halo), the hand-over level ``Ld`` and the halo width ``W = 4 * 2^Ld``
        -- or None when the mesh does not allow it (non-uniform elements, strips narrower than the halo, sizes that are
        not multiples of 2^Ld).  Strip boundaries are multiples of 2^Ld so that every level coarsens exactly as on one GPU."""
        NX, NY = self._NX, self._NY
        lx = self._grid['dx_col']   # (one element height by construction)
        if np.max(np.abs(lx - lx[0])) > 1e-12 * abs(lx[0]):
            return None
        cost = self._column_cost()

        def boundaries(Ld):
            """strip boundaries (element columns, multiples of 2^Ld): equal cost per strip, every strip at least one halo wide"""
            al, W = 1 << Ld, 4 << Ld
            units = NX // al
            need = max(1, (W if nranks > 1 else al) // al)      # units a strip must own: its neighbours' halos lie inside it
            if units < need * nranks:
                return None
            cu = np.concatenate(([0.], np.cumsum(cost.reshape(units, al).sum(axis=1))))
            cols = [0]
            for r in range(1, nranks):
                k = int(np.searchsorted(cu, cu[-1] * r / nranks, side='left'))
                if k > 0 and abs(cu[k - 1] - cu[-1] * r / nranks) <= abs(cu[k] - cu[-1] * r / nranks):
                    k -= 1
                k = min(max(k, cols[-1] + need), units - need * (nranks - r))
                cols.append(k)
            cols.append(units)
            return [k * al for k in cols]

        # Hand-over level: the one with the cheapest slowest strip.  Model of a strip's load step in units of one analytic
        # element: its owned columns by cost, its halo columns (operator, smoother and transfers pass over them, the sweep
        # does not) and the replicated coarse hierarchy (levels >= Ld of the GLOBAL grid, 4/3 of the nodes of level Ld).
        best = None
        for Ld in ((coarse_level,) if coarse_level else (4, 3, 2, 1)):
            al, W = 1 << Ld, 4 << Ld
            if NX % al or NY % al:
                continue
            gx, gy = NX >> Ld, NY >> Ld
            if gx % 2 or gy % 2 or gx * gy <= 4:      # the replicated coarse grid needs a hierarchy of its own
                continue
            cols = boundaries(Ld)
            if cols is None or (nranks > 1 and min(b - a for a, b in zip(cols[:-1], cols[1:])) < W):
                continue
            cu = np.concatenate(([0.], np.cumsum(cost)))
            t = max(cu[cols[r + 1]] - cu[cols[r]] + NY * W * ((r > 0) + (r < nranks - 1)) for r in range(nranks))
            t += (4. / 3.) * gx * gy if nranks > 1 else 0.
            if best is None or t < best[0]:
                best = (t, Ld, cols)
        if best is not None:
            _, Ld, cols = best
            W = 4 << Ld
            c0, c1 = cols[rank], cols[rank + 1]
            g0 = c0 - (W if rank > 0 else 0)
            g1 = c1 + (W if rank < nranks - 1 else 0)
            return dict(c0=c0, c1=c1, g0=g0, g1=g1, Ld=Ld, W=W, rank=rank, nranks=nranks)
        return None

    def _column_cost(self):
        """Relative cost of one load step per element column, the weights of the strip boundaries: an element of an
        analytic material counts 1 (streaming sweep + its share of the solve), an element with an SVC yield function
        ``nsv * nfeat / 8`` (its plastic corrector evaluates the support-vector sums 50 x per sweep; measured on config 4:
        2.2 us per 1585 x 6 element update against 1e-4 us for a Hill element in the streaming sweep -- whatever the exact
        ratio, strips have to balance the SVC elements first).  ``Model.strip_weights`` (array of NX column weights)
        overrides it."""
        w = getattr(self, 'strip_weights', None)
        if w is not None:
            w = np.asarray(w, dtype=float)
            if w.shape != (self._NX,) or not np.all(w > 0.):
                raise ValueError('strip_weights: NX positive column weights expected')
            return w
        if getattr(self, 'mat', None) is None or getattr(self, '_mat_id', None) is None:
            return np.full(self._NX, float(self._NY))
        per_mat = np.array([(len(m.svc['dual']) * m.Ndof / 8.) if getattr(m, 'ML_yf', False) else 1. for m in self.mat])
        if self._grid.get('mat_col') is not None:
            return per_mat[self._grid['mat_col']] * float(self._NY)
        return per_mat[self._mat_id].reshape(self._NX, self._NY).sum(axis=1)

    def strip_range(self, rank, nranks):
        """Owned element range of x-strip ``rank``: whole element columns, balanced."""
        NX, NY = self._NX, self._NY
        c0 = (NX * rank) // nranks
        c1 = (NX * (rank + 1)) // nranks
        return c0 * NY, c1 * NY

    def _ensure_engine(self):
        # content digests, not object identities or setter counters: an attribute edited in place (m.hill[0] = ...) is
        # honoured like in the reference, and a new Material at a recycled address is never mistaken for the old one
        vers = tuple(m._content_key(self._element_CV(m), parameters_only=True) for m in self.mat)
        if self._engine is not None and self._mat_versions == vers:
            return self._engine
        if self._engine is not None and self.u is not None:
            raise RuntimeError('materials were modified after the model was solved; call mesh() again')
        self._drop_engine()
        for m in self.mat:
            if m.sy is not None:
                m._no_flow_rule()  # Tresca / Barlat have no normal in the reference either (material.py:822-825)
        eng = _lib.Context(self.device)
        # one engine material per distinct Material object: a laminate lists the same material for several sections
        # (assign([A, B, A, B, A])), and the library runs only ONE 6-feature SVC material on its wave-per-element kernels
        uniq, remap = [], []
        for m in self.mat:
            j = next((k for k, q in enumerate(uniq) if q is m), None)
            if j is None:
                uniq.append(m)
                j = len(uniq) - 1
            remap.append(j)
        eng.set_materials([m._record(self._element_CV(m)) for m in uniq])
        self._eng_uniq = uniq
        eng.set_wh_mode(self.wh_carry != 'per_point')
        remap = np.asarray(remap, dtype=np.int64)
        g = self._grid
        eng_mat_col = None if g['mat_col'] is None else remap[g['mat_col']]
        eng_mat_el = None if g['mat_el'] is None else remap[g['mat_el']]
        e0, e1 = 0, self.Nel
        self._strip = None
        plan = None
        if self._shard is not None:
            rank, nranks, uid = self._shard
            if uid is not None:
                eng.comm_init(uid, rank, nranks)
            elif getattr(self, '_host_allreduce', None) is not None:
                eng.comm_init_callback(rank, nranks, self._host_allreduce)
            mode = getattr(self, '_dist_mode', None)
            if mode != 'replicated' and self.operator != 0 and self.precond != 0:
                plan = self.strip_plan(rank, nranks, getattr(self, '_strip_level', None))
            if plan is None and mode == 'strip':
                raise ValueError('distribute: the strip-local engine needs a uniform structured grid whose strips are at '
                                 'least one halo (4 * 2^level columns) wide, with NX, NY multiples of 2^level')
            e0, e1 = self.strip_range(rank, nranks)
        if plan is not None:
            # this rank's strip (owned columns + halo) as a standalone local mesh; same numbering rules (model.py:893, 935)
            NY, nyn = self._NY, self._NY + 1
            g0, g1 = plan['g0'], plan['g1']
            nxl = g1 - g0
            nnode_l = (nxl + 1) * nyn
            # material state and sweep on the owned columns only; the stiffness generators of the halo columns come from
            # the neighbours that own them (PLFX_STRIP_SWEEP_HALO=1: sweep the halo elements redundantly instead)
            lean = os.environ.get('PLFX_STRIP_SWEEP_HALO', '0') != '1'
            eo0, eo1 = ((plan['c0'] - g0) * NY, (plan['c1'] - g0) * NY) if lean else (0, nxl * NY)
            eng.set_mesh_structured(nxl, NY, g['dx_col'][g0:g1], g['dy'], self.thick, self.planestress,
                                    mat_col=None if eng_mat_col is None else eng_mat_col[g0:g1],
                                    mat_el=None if eng_mat_el is None else eng_mat_el[g0 * NY:g1 * NY], el_begin=eo0, el_end=eo1)
            eng.set_grid(nxl, NY)
            eng.set_strip(plan['c0'] - g0, plan['c1'] - g0, g0, self._NX, plan['Ld'])
            last = plan['rank'] == plan['nranks'] - 1
            plan.update(node0=g0 * nyn, nnode=nnode_l, el0=g0 * NY, nel=nxl * NY, state_el0=g0 * NY + eo0,
                        own_nodes=(plan['c0'] * nyn, (self._NX + 1 if last else plan['c1']) * nyn))
            self._strip = plan
            e0, e1 = plan['c0'] * NY, plan['c1'] * NY
        elif getattr(self, '_explicit_mesh', False):   # the index arrays handed over as arrays (plfx_set_mesh): same engine state
            emid = remap[self._mat_id]
            eng.set_mesh(self._conn, emid, self._lxy, self.Nnode, self.thick, self.planestress, e0, e1)
            eng.set_grid(self._NX, self._NY)
        else:
            eng.set_mesh_structured(self._NX, self._NY, g['dx_col'], g['dy'], self.thick, self.planestress, mat_col=eng_mat_col,
                                    mat_el=eng_mat_el, el_begin=e0, el_end=e1)
            eng.set_grid(self._NX, self._NY)  # structured numbering -> multigrid preconditioner where possible
        if self.precond is not None:
            eng.set_precond(self.precond)
        if self.operator is not None:
            eng.set_operator(self.operator)
        self._e0, self._e1 = e0, e1
        if self.Nel >= 65536 and eng.precond_info()[0] != 1 and self.precond is None:
            # correct but orders of magnitude slower: say so instead of silently falling back
            warnings.warn('mesh of {} elements without a structured grid hierarchy (explicit element sizes, column widths more '
                          'than 1.5 apart, or a grid too irregular to coarsen): the solve falls back to Jacobi-PCG on the assembled '
                          'operator'.format(self.Nel))
        self._dev_coll = eng.comm_info()[2]  # RCCL communicator: scalars are all-reduced inside the library
        self._engine = eng
        self._mat_versions = vers
        plastic = any(m.sy is not None for m in self.mat)
        if plastic != self.nonlin:
            raise RuntimeError('plasticity of a material changed after assign(); call assign() again')
        return eng

    def _state(self, name):
        """Element results downloaded from HBM on first access after a solve."""
        if name not in self._cache:
            ids = {'sig': _lib.ST_SIG, 'eps': _lib.ST_EPS, 'epl': _lib.ST_EPL, 'elstiff': _lib.ST_ELSTIFF,
                   'res_sig': _lib.ST_RES_SIG, 'res_depl': _lib.ST_RES_DEPL, 'max_steps': _lib.ST_MAXSTEPS,
                   'fyn': _lib.ST_FYN}
            a = self._ensure_engine().state_get(ids[name])
            if self._strip is not None:  # local strip: the owned part of the state arrays into a full-size array
                full = np.zeros((self.Nel,) + a.shape[1:])
                o = self._e0 - self._strip['state_el0']
                full[self._e0:self._e1] = a[o:o + self._e1 - self._e0]
                a = full
            elif self._shard is not None:  # place the owned strip into a full-size array
                full = np.zeros((self.Nel,) + a.shape[1:])
                full[self._e0:self._e1] = a
                a = full
            self._cache[name] = a
        return self._cache[name]

    # ------------------------------------------------------------------ assembly
    def setupK(self):
        """Assemble the system stiffness matrix (model.py:954-977).  Returned as a
        ``scipy.sparse.csr_matrix`` (the reference returns the same matrix as a dense array)."""
        eng = self._ensure_engine()
        eng.assemble()
        return eng.get_csr()

    # ------------------------------------------------------------------ boundary conditions
    def _bc_plan(self):
        """Index structure of calc_BC (model.py:1070-1206): depends on the BC flags and node sets only, so it is built
        once and reused by every solve.  Displacement segments in the reference's order left, bottom, right, top,
        node set (x before y); force segments with the per-node share of the edge force."""
        key = (tuple(self.ubcleft), tuple(self.ubcbot), tuple(self.ubcright), tuple(self.ubctop),
               tuple(self.ubcn) if self.noset is not None else None,
               None if self.noset is None else tuple(self.noset), self.Nnode,
               None if self._strip is None else (self._strip['node0'], self._strip['nnode']))
        st = self._bc_struct
        if st is not None and st[0] == key:
            return st[1]
        dseg, fseg = [], []   # (source, k, dof indices[, shares])

        def share(nodes, npart, pos, length):
            hh = np.full(len(nodes), 1. / (npart - 1))   # share of the edge force per node
            hp = self._node_coord(nodes, pos)
            hh[(hp < 1.e-3) | (hp > length - 1.e-3)] *= 0.5  # half on corner nodes
            return hh

        strip = self._strip   # strip-local engine: only the nodes of this rank's local grid, in its local numbering

        def sel(nodes):
            a = np.asarray(nodes, dtype=np.int64)
            if strip is None:
                return a
            return a[(a >= strip['node0']) & (a < strip['node0'] + strip['nnode'])]

        nl, nb, nr, nt = sel(self.noleft), sel(self.nobot), sel(self.noright), sel(self.notop)
        for k in range(2):
            if self.ubcleft[k]:
                dseg.append(('l', k, 2 * nl + k))
        for k in range(2):
            if self.ubcbot[k]:
                dseg.append(('b', k, 2 * nb + k))
        for k in range(2):
            if self.ubcright[k]:
                dseg.append(('r', k, 2 * nr + k))
            else:
                fseg.append(('r', k, 2 * nr + k, share(nr, self.NnodeY, 1, self.leny)))
        for k in range(2):
            if self.ubctop[k]:
                dseg.append(('t', k, 2 * nt + k))
            else:
                fseg.append(('t', k, 2 * nt + k, share(nt, self.NnodeX, 0, self.lenx)))
        if self.noset is not None:
            ns = sel(self.noset)
            for k in range(2):
                if self.ubcn[k]:
                    dseg.append(('n', k, 2 * ns + k))
                else:
                    fseg.append(('n', k, 2 * ns + k, np.ones(len(ns))))
        if strip is not None:   # global -> local DOF numbers (node id = column * NnodeY + row on both grids)
            off = 2 * strip['node0']
            dseg = [(src, k, idx - off) for src, k, idx in dseg]
            fseg = [(src, k, idx - off, hh) for src, k, idx, hh in fseg]
        plan = {'dseg': [(src, k) for src, k, _ in dseg], 'fseg': fseg}
        if dseg:
            idx = np.concatenate([d[2] for d in dseg])
            plan['seg'] = np.concatenate([np.full(len(d[2]), i, dtype=np.intp) for i, d in enumerate(dseg)])
            presc, first_pos, inv = np.unique(idx, return_index=True, return_inverse=True)
            plan.update(idx=idx, presc=presc.astype(np.int32), first_pos=first_pos, inv=inv)
            # pairs of entries on the same DOF (corner nodes): only these can be inconsistent
            plan['dup'] = np.nonzero(first_pos[inv] != np.arange(len(idx)))[0]
        self._bc_struct = (key, plan)
        return plan

    def _bc_data(self, bcl0, bcb0, dbcr, dbct, dbcn):
        """calc_BC (model.py:1070-1206) as data, in O(boundary) work: prescribed DOFs (ascending),
        value written to du (first application), multiplicity-weighted value for the right-hand side
        (a DOF shared by two edges enters the rhs twice, :1115-1122, 1163-1170), external forces."""
        if self.noset is not None and dbcn is None:
            raise ValueError('No BC for selected node set given.')
        plan = self._bc_plan()
        src = {'l': bcl0, 'b': bcb0, 'r': dbcr, 't': dbct, 'n': dbcn}
        if plan['dseg']:
            val = np.array([float(src[s][k]) for s, k in plan['dseg']])[plan['seg']]
            presc, first_pos, inv = plan['presc'], plan['first_pos'], plan['inv']
            first = val[first_pos]
            w = np.bincount(inv, weights=val, minlength=len(presc))
            dup = plan['dup']
            if len(dup):
                bad = val[dup] != first[inv[dup]]
                if np.any(bad):
                    j = dup[bad][0]
                    warnings.warn('Inconsistent BC at DOF {} ({} vs {}).'.format(plan['idx'][j], first[inv[j]], val[j]))
        else:
            presc = np.zeros(0, dtype=np.int32)
            first = w = np.zeros(0)
        fext = None
        for s, k, fidx, hh in plan['fseg']:
            v = float(src[s][k])
            if v != 0.:
                if fext is None:
                    fext = np.zeros(self._ndof_local())
                np.add.at(fext, fidx, v * hh)
        return presc, first, w, fext

    def _nodal(self, a):
        """strip-local engine: the local nodal array (owned + halo columns) placed into a full-size one"""
        if self._strip is None:
            return a
        full = np.zeros(self.Ndof)
        o = 2 * self._strip['node0']
        full[o:o + len(a)] = a
        return full

    def _ndof_local(self):
        return self.Ndof if self._strip is None else 2 * self._strip['nnode']

    def free_dofs(self):
        """The reference's ``ind`` list (ascending free DOFs) for the current BC flags."""
        z = np.zeros(2)
        presc = self._bc_data(z, z, z, z, z if self.noset is not None else None)[0]
        mask = np.ones(self.Ndof, dtype=bool)
        mask[presc] = False
        return np.nonzero(mask)[0]

    def _bc_register(self, eng):
        """hand calc_BC's index structure to the library once per solve(); later calls pass one value per segment"""
        plan = self._bc_plan()
        if self._bc_registered is not plan:
            if plan['dseg']:
                segs = np.bincount(plan['seg'], minlength=len(plan['dseg']))
                eng.set_bc_plan(segs, plan['idx'])
            else:
                eng.set_bc_plan(np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.int32))
            code = {'l': 0, 'b': 1, 'r': 2, 't': 3, 'n': 4}
            fseg = plan['fseg']
            eng.set_bc_sources([code[s] for s, _ in plan['dseg']], [k for _, k in plan['dseg']],
                               [code[f[0]] for f in fseg], [f[1] for f in fseg], [len(f[2]) for f in fseg],
                               np.concatenate([f[2] for f in fseg]) if fseg else np.zeros(0, dtype=np.int32),
                               np.concatenate([f[3] for f in fseg]) if fseg else np.zeros(0))
            self._bc_registered = plan
        return plan

    def _bc_apply(self, eng, bcl0, bcb0, dbcr, dbct, dbcn):
        """calc_BC (model.py:1070-1206) through the registered plan: segment values and (rarely) external forces"""
        if self.noset is not None and dbcn is None:
            raise ValueError('No BC for selected node set given.')
        plan = self._bc_register(eng)
        src = {'l': bcl0, 'b': bcb0, 'r': dbcr, 't': dbct, 'n': dbcn}
        vals = [float(src[s][k]) for s, k in plan['dseg']]
        fext = None
        for s, k, fidx, hh in plan['fseg']:
            v = float(src[s][k])
            if v != 0.:
                if fext is None:
                    fext = np.zeros(self._ndof_local())
                np.add.at(fext, fidx, v * hh)
        bad = eng.apply_bc_plan(vals, fext)
        if bad >= 0:
            j = bad
            first = vals[plan['seg'][plan['first_pos'][plan['inv'][j]]]]
            warnings.warn('Inconsistent BC at DOF {} ({} vs {}).'.format(plan['idx'][j], first, vals[plan['seg'][j]]))

    def _solve_lin(self, eng, bc, warm):
        self._bc_apply(eng, *bc)
        it, rr, ok = eng.solve(self.cg_rtol, self.cg_maxit, warm)
        self.solver_stats.append((it, rr))
        if not ok:
            warnings.warn('PCG reached the iteration limit (relres={:.2e})'.format(rr))

    def _calc_scf(self, eng, sld):
        """Load-step scaling factor (model.py:1036-1067) from per-element values reduced on the GPU."""
        if self._shard is None or self._dev_coll:  # sharded with RCCL: the library all-reduces the statistics itself
            cnt, mn, s, s2 = eng.scf_all(sld)
            if cnt == 0:
                return 1.
            mean = s / cnt
        else:  # the mean is global: one collective between the two passes
            cnt, mn, s = eng.scf_stats(sld)
            cnt, mn, s = self._allreduce_scf(cnt, mn, s)
            if cnt == 0:
                return 1.
            mean = s / cnt
            s2 = self._allreduce_sum(eng.scf_sumsq(mean))
        std = np.sqrt(s2 / cnt)
        if std < 0.1:
            scf = mn
        else:
            scf = np.maximum(1.e-3, mean - std)
        if scf < 1.e-3:
            scf = 1.e-3
        return float(scf)

    # collectives of the host-side scalars of the Python load-step driver (sharded runs whose library communicator is the
    # host-staged transport): through the library's own communicator (plfx_allreduce_host) -- the package itself needs no
    # process-group library; sums / minima are identical on every rank
    def _host_reduce(self, values, op):
        """all-reduce <= 32 host doubles (op 0 sum, 3 min): through the engine's communicator, or -- before an engine
        exists -- through the host-transport callback given to ``distribute``"""
        v = np.ascontiguousarray(values, dtype=np.float64)
        if self._engine is not None:
            return self._engine.allreduce_host(v, op=op)
        fn = getattr(self, '_host_allreduce', None)
        if fn is None:
            raise RuntimeError('sharded model without a communicator: call distribute(rank, nranks, uid | host_allreduce=...)')
        out = v.copy()
        fn(out, op)
        return out

    def _allreduce_sum(self, x):
        v = np.atleast_1d(np.asarray(x, dtype=np.float64))
        out = np.empty_like(v)
        for i in range(0, len(v), 32):           # plfx_allreduce_host takes <= 32 doubles per call
            out[i:i + 32] = self._host_reduce(v[i:i + 32], 0)
        return out if np.ndim(x) else float(out[0])

    def _allreduce_scf(self, cnt, mn, s):
        g = self._host_reduce(np.array([cnt, s], dtype=np.float64), 0)
        m = self._host_reduce(np.array([mn], dtype=np.float64), 3)
        return int(round(g[0])), float(m[0]), float(g[1])

    def _allreduce_flags(self, change, conv):
        g = self._host_reduce(np.array([float(change), float(not conv)]), 0)
        return bool(g[0] > 0.), not bool(g[1] > 0.)

    # ------------------------------------------------------------------ solution
    def solve(self, min_step=None, verb=False):
        """Solve the (non-linear) boundary-value problem (model.py:979-1450); see `_solve_steps`.

        The cyclic garbage collector is paused for the duration of the load-step loop: the loop allocates no reference
        cycles, while a full collection of a process holding a large mesh (node / element lists of ~1e6 entries) stalls
        the loop for milliseconds -- as long as a whole load step on the GPU."""
        import gc
        was_enabled = gc.isenabled()
        gc.disable()
        try:
            return self._solve_steps(min_step=min_step, verb=verb)
        finally:
            if was_enabled:
                gc.enable()

    def _solve_steps(self, min_step=None, verb=False):
        """Solve the (non-linear) boundary-value problem (model.py:979-1450).

        Same load-step control as the reference: elastic predictor, load-step scaling ``calc_scf``
        (first 10 steps), up to 16 stiffness iterations with halving of the increment (first 6 steps),
        state update, homogenisation.  Results: ``u, f, sgl, egl, epgl, glob, nsteps, niter,
        co_nconv`` and ``element[i].sig/eps/epl``."""
        if self.Nnode is None:
            raise AttributeError('Attributes for mesh not set, but required by solver.')
        eng = self._ensure_engine()
        eng.set_wh_mode(self.wh_carry != 'per_point')
        self._cache = {}
        n_stats0 = len(self.solver_stats)
        dim = 2
        first_call = self.u is None
        if first_call:
            eng.state_reset()
            self.sgl, self.egl, self.epgl = (np.zeros((1, 6)) for _ in range(3))
            for name in ('bcr_mem', 'bct_mem') + (('bcn_mem',) if self.noset is not None else ()):
                setattr(self, name, np.zeros(dim))
        # boundary values reached so far; a second call resumes from the previous solution (model.py:1235-1239)
        bcr0, bct0 = ((np.zeros(dim), np.zeros(dim)) if first_call else (self.bcr_mem, self.bct_mem))
        if self.noset is not None:
            bcn0 = np.zeros(dim) if first_call else self.bcn_mem
        bcl0, bcb0 = self.bcl, self.bcb
        sgl, egl, epgl = list(self.sgl), list(self.egl), list(self.epgl)
        wh = [(k, m) for k, m in enumerate(getattr(self, '_eng_uniq', [])) if getattr(m, 'whdat', False) and m.ML_yf]
        for k, m in wh:   # the value the Material object holds NOW enters the first response() call (material.py:808-814)
            eng.wh_carry(k, m.khard)
        self._bc_registered = None
        self._finish_register(eng)
        eng.assemble()
        # loading direction for the ML yield-point search (model.py:1245-1258)
        sld = np.zeros(6)
        for bc, k, comp in ((self.bcr, 0, 0), (self.bct, 1, 1), (self.bcr, 1, 5), (self.bct, 0, 5)):  # later entries win
            if abs(bc[k]) > 1.e-6:
                sld[comp] = np.sign(bc[k])
        if not sld.any():
            warnings.warn('solve: inconsistent BC sld={}, bct={}, bcr={}'.format(sld, self.bct, self.bcr))
            sld[0] = 1.
        il = 0
        nit = 0
        niter = []
        co_nconv = []
        bc_inc = True
        pending = None   # slot of a load step whose end-of-step data are still to be collected
        defer = self._defer_finish and eng.lib_has_mailbox()
        # (single GPU: sgl / egl / epgl / glob are filled in after the loop; with strips the boundary sums are a collective per step)
        lazy = [] if (defer and self._strip is None and self._shard is None and not verb) else None

        def record_global(fin_data):
            self._calc_global_device(eng, fin_data)
            sgl.append(self.glob['sig'])
            egl.append(self.glob['eps'])
            epgl.append(self.glob['epl'])
        nconv = 0
        warm = not first_call
        dbcn = None
        while bc_inc:
            # what is left of the boundary values, spread over the load steps still to go (model.py:1262-1285)
            togo = 1 if min_step is None else max(1, min_step - il)
            dbcr = max_dbcr = (self.bcr - bcr0) / togo
            dbct = max_dbct = (self.bct - bct0) / togo
            if self.noset is not None:
                dbcn = max_dbcn = (self.bcn - bcn0) / togo  # ONE array under two names, exactly as in the reference (model.py:1285)
            native = self._native_step and not verb and (self._shard is None or self._dev_coll)
            if native:
                # the body of the load step runs inside the library (plfx_load_step): predictor, calc_scf, stiffness
                # iterations, state update; verb=True keeps the Python transcription below for its trace output
                st = self._step_io
                st.il, st.nonlin, st.warm = il, int(bool(self.nonlin)), int(bool(warm))
                st.has_nodeset = int(self.noset is not None)
                st.maxit, st.rtol = int(self.cg_maxit), float(self.cg_rtol)
                st.bcl0[:], st.bcb0[:] = [float(v) for v in bcl0], [float(v) for v in bcb0]
                st.max_dbcr[:], st.max_dbct[:] = [float(v) for v in max_dbcr], [float(v) for v in max_dbct]
                st.bcr[:], st.bct[:] = [float(v) for v in self.bcr], [float(v) for v in self.bct]
                st.bcr0[:], st.bct0[:] = [float(v) for v in bcr0], [float(v) for v in bct0]
                if self.noset is not None:
                    st.max_dbcn[:], st.bcn[:], st.bcn0[:] = ([float(v) for v in max_dbcn], [float(v) for v in self.bcn],
                                                              [float(v) for v in bcn0])
                st.sld[:] = [float(v) for v in sld]
                self._bc_register(eng)
                # end-of-step data (boundary u, f and the element sums of calc_global) are not waited for: the library
                # posts them into one of two pinned slots and this loop collects them one step later, so its own
                # bookkeeping and the next predictor overlap the state update on the GPU
                st.defer_slot = (il & 1) + 1 if defer else 0
                fin = eng.load_step(st)
                warm = True
                dbcr, dbct = np.array(st.dbcr[:]), np.array(st.dbct[:])
                if self.noset is not None:
                    dbcn = np.array(st.dbcn[:])
                nit = st.nit
                nconv += st.nconv
                self.n_sweeps += st.nsweeps
                for q in range(min(st.nsolves, 40)):
                    self.solver_stats.append((st.its[q], st.relres[q]))
                if st.soft_fail:
                    warnings.warn('PCG reached the iteration limit in {} solve(s) of load step {}'.format(st.soft_fail, il))
                if st.inconsistent_entry >= 0:
                    warnings.warn('Inconsistent BC at DOF {}.'.format(self._bc_plan()['idx'][st.inconsistent_entry]))
            # elastic predictor with the stiffness of the previous step (model.py:1290-1291)
            if not native:
                self._solve_lin(eng, (bcl0, bcb0, dbcr, dbct, dbcn), warm)
                warm = True
            if self.nonlin and not native:
                scale_bc = self._calc_scf(eng, sld) if il < 10 else 1.
                dbcr = max_dbcr * scale_bc
                dbct = max_dbct * scale_bc
                nit = 0
                change = True
                conv = False
                if verb:
                    print('***Load step #', il)
                    print('scaling factor', scale_bc)
                while (change or not conv) and nit <= 15:
                    if il < 6 and nit > 1:
                        # reduce the load increment to reach convergence (model.py:1308-1330)
                        # halve it, but stay between 5 % of the planned increment and what is left to the target value
                        edges = [(max_dbcr, self.bcr, bcr0, dbcr), (max_dbct, self.bct, bct0, dbct)]
                        if self.noset is not None:
                            edges.append((max_dbcn, self.bcn, bcn0, dbcn))
                        for k in range(dim):
                            for (mx, tot, cur0, d) in edges:
                                inner, outer = (np.minimum, np.maximum) if mx[k] >= 0 else (np.maximum, np.minimum)
                                d[k] = outer(0.05 * mx[k], inner(tot[k] - cur0[k], d[k] * 0.5))
                    eng.assemble()  # updated tangent stiffness (model.py:1333)
                    self._solve_lin(eng, (bcl0, bcb0, dbcr, dbct, dbcn), True)
                    change, conv = eng.sweep(nit)  # material response of every element (model.py:1340-1361)
                    self.n_sweeps += 1
                    if self._shard is not None and not self._dev_coll:
                        change, conv = self._allreduce_flags(change, conv)
                    if verb:
                        if not conv:
                            print('\n  ###  Warning: No convergence of plasticity algorithm in trial step #', nit)
                        print('+++Inner trial step #', nit)
                        print('load increment right:', dbcr)
                        print('load increment top:', dbct)
                        if self.noset is not None:
                            print('load increment set:', dbcn)
                    if not conv:
                        nconv += 1
                    nit += 1
            # update internal variables with the results of the load step (model.py:1383-1392)
            if not native:
                fin = eng.finish_step()  # update_state + boundary u, f + element sums: one call, one synchronisation
            il += 1
            niter.append(nit - 1)
            co_nconv.append(nconv)
            # another load step while any non-zero boundary value is not reached yet (model.py:1395-1411)
            reached = [(bcr0, dbcr, self.bcr), (bct0, dbct, self.bct)]
            if self.noset is not None:
                reached.append((bcn0, dbcn, self.bcn))
            bc_inc = False
            for cur0, d, tot in reached:
                cur0 += d
                bc_inc = bc_inc or bool(np.any((np.abs(cur0 - tot) > 1.e-6) & (np.abs(tot) > 1.e-9)))
            if native and defer:
                if pending is not None:
                    if lazy is not None:
                        # homogenisation of the step before: its data are taken out of the pinned slot now (the slot is reused),
                        # the numpy work on them (two bincounts over the boundary nodes, ~40 us) waits until the loop is done --
                        # between two plfx_load_step calls the GPU only has the tail of the state update left to hide host time
                        lazy.append(tuple(a.copy() for a in eng.finish_fetch(pending)))
                    else:
                        record_global(eng.finish_fetch(pending))
                pending = (il - 1) & 1
            else:
                record_global(fin)
            if self._step_hook is not None:
                self._step_hook(il)
            if self._max_load_steps is not None and il >= self._max_load_steps:
                bc_inc = False
            if verb:
                print('Iteration step #', nit)
                print('Load increment ', il, 'total', self.ubctop, 'top ', bct0, '/', self.bct, '; last step ', dbct)
                print('Load increment ', il, 'total', self.ubcright, 'rhs', bcr0, '/', self.bcr, '; last step ', dbcr)
                print('Global strain: ', np.around(self.glob['eps'], decimals=5))
                print('Global stress: ', np.around(self.glob['sig'], decimals=3))
                print('Global plastic strain: ', np.around(self.glob['epl'], decimals=6))
                print('----------------------------')
        for fin_data in (lazy or ()):
            record_global(fin_data)
        if pending is not None:
            record_global(eng.finish_fetch(pending))
        self.sgl, self.egl, self.epgl = np.array(sgl), np.array(egl), np.array(epgl)
        self.bct_mem = bct0
        self.bcr_mem = bcr0
        if self.noset is not None:
            self.bcn_mem = bcn0
        self.nsteps = il
        self.niter = niter
        self.co_nconv = co_nconv
        if wh and eng.wh_info()[0]:   # ... and the objects keep what the last gradient evaluation of the run left
            for k, m in wh:
                m.khard = eng.wh_carry(k)
            if eng.wh_unresolved:
                warnings.warn('{} sweep(s) ended with an unsettled hardening-modulus chain (pass cap of the library reached: PLFX_WH_MAXPASS, default 512)'.format(eng.wh_unresolved))
        # the reference's LU always returns; an iterative solve can end above its tolerance (nearly singular tangents at a
        # limit load): say so instead of continuing silently
        bad = [r for (_, r) in self.solver_stats[n_stats0:] if not r <= 10. * self.cg_rtol]
        if bad:
            warnings.warn('{} of {} linear solves ended above the tolerance (worst relative residual {:.2e}, rtol {:g})'
                          .format(len(bad), len(self.solver_stats) - n_stats0, max(bad), self.cg_rtol))
        self.u = self._nodal(eng.state_get(_lib.ST_U))
        self.f = self._nodal(eng.state_get(_lib.ST_F))
        self.du = self._nodal(eng.state_get(_lib.ST_DU))
        self._cache = {}

    # ------------------------------------------------------------------ homogenisation
    def bcval(self, nodes):
        """Average displacement and total force at a node set (model.py:1452-1471)."""
        idx = 2 * np.asarray(nodes, dtype=np.int64)
        n = len(idx)
        return (np.sum(self.u[idx]) / n, np.sum(self.u[idx + 1]) / n, np.sum(self.f[idx]), np.sum(self.f[idx + 1]))

    def _glob_from(self, bv, sums):
        (uxl, uyl, fxl, fyl), (uxr, uyr, fxr, fyr), (uxb, uyb, fxb, fyb), (uxt, uyt, fxt, fyt) = bv
        g = self.glob
        g['ebc1'] = (uxr - uxl) / self.lenx
        g['sbc1'] = 0.5 * (fxr - fxl) / (self.leny * self.thick)
        g['ebc21'] = (uyr - uyl) / self.lenx
        g['sbc21'] = 0.5 * (fyr - fyl) / (self.leny * self.thick)
        g['ebc2'] = (uyt - uyb) / self.leny
        g['sbc2'] = 0.5 * (fyt - fyb) / (self.lenx * self.thick)
        g['ebc12'] = (uxt - uxb) / self.leny
        g['sbc12'] = 0.5 * (fxt - fxb) / (self.lenx * self.thick)
        Vm = self.lenx * self.leny * self.thick
        g['sig'] = sums[0] / Vm
        g['eps'] = sums[1] / Vm
        g['epl'] = sums[2] / Vm

    def _finish_register(self, eng):
        sets = (self.noleft, self.noright, self.nobot, self.notop)
        if self._bnd_idx is None:
            parts, seg = [], []
            for q, nodes in enumerate(sets):
                a = np.asarray(nodes, dtype=np.int64)
                if self._strip is not None:  # this rank's owned nodes only (disjoint over the ranks), local numbering
                    lo, hi = self._strip['own_nodes']
                    a = a[(a >= lo) & (a < hi)] - self._strip['node0']
                idx = 2 * a
                parts.extend((idx, idx + 1))
                seg.extend((np.full(len(idx), 2 * q), np.full(len(idx), 2 * q + 1)))
            self._bnd_idx = np.concatenate(parts)
            # [x of set 0, y of set 0, x of set 1, ...]: 8 segment sums; the divisor is the size of the WHOLE node set
            self._bnd_offs = (np.concatenate(seg).astype(np.intp), np.array([len(nodes) for nodes in sets], dtype=float))
        eng.set_finish_set(self._bnd_idx)

    def _calc_global_device(self, eng, fin=None):
        """calc_global during solve: boundary DOFs gathered from HBM, element sums reduced on the GPU."""
        if fin is None:
            self._finish_register(eng)
            uu = eng.gather(_lib.ST_U, self._bnd_idx)
            ff = eng.gather(_lib.ST_F, self._bnd_idx)
            sums = eng.global_sums()
        else:
            uu, ff, sums = fin
        if self._bnd_offs is None:
            self._bnd_idx = None
            self._finish_register(eng)
        seg, cnt = self._bnd_offs
        su = np.bincount(seg, weights=uu, minlength=8)
        sf = np.bincount(seg, weights=ff, minlength=8)
        if self._strip is not None:   # boundary sums over the strips (plfx_allreduce_host); element sums arrive all-reduced
            tot = eng.allreduce_host(np.concatenate((su, sf)))
            su, sf = tot[:8], tot[8:]
        bv = [(su[2 * q] / cnt[q], su[2 * q + 1] / cnt[q], sf[2 * q], sf[2 * q + 1]) for q in range(4)]
        if self._shard is not None and not (self._dev_coll and fin is not None):
            sums = self._allreduce_sum(sums.ravel()).reshape(3, 6)
        self._glob_from(bv, sums)

    def calc_global(self):
        """Global quantities from boundary nodes and element averages (model.py:1473-1511)."""
        eng = self._ensure_engine()
        if self._strip is not None and self.u is not None:
            return   # the homogenised values of the last load step are current (element sums over strips need the halo masks)
        self._calc_global_device(eng)

    def plot(self, *args, **kw):
        raise NotImplementedError('plotting is out of scope of pylabfea_amd (SURVEY.md §2); '
                                  'use the reference package on the arrays u, npos, element[i].sig')
