"""CPU tests (gloo, world size 2 and 3) of the host side of the strip-local multi-GPU engine (plfx_set_strip, DESIGN.md
section 6): the strip plan of ``Model.strip_plan``, the validity-width rule the library enforces, the host-staged transport
(``pylabfea_amd.host_transport``: sums, minima and the neighbour exchange, op 100) and the two identities the engine
relies on, checked with the CPU oracle's assembly:

  * after the halo refresh (to the left neighbour my node columns c0+1..c0+W, to the right one c1-W..c1-1) every local
    column of a node vector carries the global values;
  * the operator assembled from the LOCAL elements only (owned + halo columns), applied to that vector, equals the global
    K p on every local column except the outermost one of an interior edge (whose rows miss the elements beyond it)."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import pylabfea_amd as FE


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def make_model(NX, NY):
    mat = FE.Material()
    mat.elasticity(E=200.e3, nu=0.3)
    mat.plasticity(sy=100., hill=[0.7, 1., 1.4, 1., 1.2, 0.8], khard=100., sdim=6)
    fe = FE.Model(dim=2)
    fe.geom([4. * NX / NY], LY=4.)
    fe.assign([mat])
    fe.bcleft(0.)
    fe.bcbot(0.)
    fe.bcright(0., 'force')
    fe.bctop(0.002 * fe.leny, 'disp')
    fe.mesh(NX=NX, NY=NY)
    return fe


def validity(W, Ld):
    """the rule of plfx_set_strip: (v, g) = validity of the level-Ld right-hand side / of z beyond the owned columns"""
    a, v = [], W
    for l in range(Ld):
        a.append(min(v, (W >> l) - 1))
        v = (a[l] - 3) // 2 if a[l] - 3 >= 0 else -1
    g = W >> Ld
    if v >= 0:
        for l in range(Ld - 1, -1, -1):
            g = min(a[l] - 1, 2 * g) - 2
    return v, g


def test_validity_widths():
    for Ld in range(1, 7):
        v, g = validity(4 << Ld, Ld)
        assert v >= 0 and g >= 1, (Ld, v, g)            # halo of 4 * 2^Ld columns: what strip_plan chooses
        if Ld >= 3:
            v, g = validity((4 << Ld) - (1 << Ld), Ld)  # one alignment unit less is not enough on the deep hand-overs
            assert v < 0 or g < 1, (Ld, v, g)
    assert validity(32, 3) == (1, 2) and validity(64, 4) == (1, 2)


@pytest.mark.parametrize('NX,NY,nranks', [(128, 32, 2), (128, 32, 4), (192, 64, 3), (8192, 1024, 8), (4096, 1024, 4),
                                           (2048, 2048, 8)])
def test_strip_plan(NX, NY, nranks):
    fe = FE.Model(dim=2)
    fe._NX, fe._NY = NX, NY
    fe._grid = {'dx_col': np.full(NX, 0.125), 'dy': 0.125, 'mat_col': None, 'mat_el': None}   # (what Model.mesh keeps of the grid)
    plans = [fe.strip_plan(r, nranks) for r in range(nranks)]
    assert all(p is not None for p in plans)
    Ld, W = plans[0]['Ld'], plans[0]['W']
    assert W == 4 << Ld and validity(W, Ld)[1] >= 1
    assert plans[0]['c0'] == 0 and plans[-1]['c1'] == NX
    for a, b in zip(plans[:-1], plans[1:]):
        assert a['c1'] == b['c0']
    for r, p in enumerate(plans):
        assert p['Ld'] == Ld and p['c1'] - p['c0'] >= W
        assert p['g0'] == (p['c0'] - W if r > 0 else 0) and p['g1'] == (p['c1'] + W if r < nranks - 1 else NX)
        assert 0 <= p['g0'] and p['g1'] <= NX
        for q in (p['c0'], p['c1'], p['g0'], p['g1']):
            assert q % (1 << Ld) == 0                   # every level coarsens exactly as on one GPU
    if NX == 8192:
        assert Ld == 4 and W == 64                      # the bench's weak-scaling layout: 1024 owned + 64 halo columns
    if (NX, NY, nranks) == (2048, 2048, 8):
        assert Ld == 3 and W == 32                      # config 5: 256-column strips -> halo an eighth of the width


def test_strip_plan_balances_svc_columns():
    """config 5's laminate [2,1,2,1,2] (J2 | SVC | J2 | SVC | J2) on 8 strips: equal column counts would put every SVC element
    on two of the eight ranks; the boundaries follow the cost of the columns instead"""
    rng = np.random.default_rng(0)
    a = FE.Material(name='J2')
    a.elasticity(E=200.e3, nu=0.3)
    a.plasticity(sy=150., khard=500., sdim=6)
    b = FE.Material(name='ML', num=2)
    b.elasticity(E=151220., nu=0.3)
    b.plasticity(sy=46.76, sdim=6)
    b.set_svc(rng.normal(size=(800, 6)), rng.normal(size=800), 0.1, 2.5, 50.)
    fe = FE.Model(dim=2)
    fe.geom([2, 1, 2, 1, 2], LY=1.)
    fe.assign([a, b, a, b, a])
    fe.bcleft(0.)
    fe.bcbot(0.)
    fe.bcright(0., 'force')
    fe.bctop(0.001, 'disp')
    fe.mesh(NX=512, NY=64)
    plans = [fe.strip_plan(r, 8) for r in range(8)]
    assert all(p is not None for p in plans)
    W = plans[0]['W']
    svc_col = np.zeros(512, dtype=bool)
    svc_col[128:192] = svc_col[320:384] = True
    n_svc = [int(svc_col[p['c0']:p['c1']].sum()) for p in plans]
    assert sum(n_svc) == 128 and max(n_svc) <= 24 and min(n_svc) >= 8, n_svc      # 16 per rank were perfect
    assert min(p['c1'] - p['c0'] for p in plans) >= W
    for x, y in zip(plans[:-1], plans[1:]):
        assert x['c1'] == y['c0']
    # explicit weights win over the material model
    fe.strip_weights = np.ones(512)
    assert [fe.strip_plan(r, 8)['c0'] for r in range(8)] == [64 * r for r in range(8)]
    fe.strip_weights = np.ones(5)
    with pytest.raises(ValueError):
        fe.strip_plan(0, 8)


def test_strip_plan_refuses_what_cannot_work():
    fe = FE.Model(dim=2)
    fe._NX, fe._NY = 96, 24
    fe._grid = {'dx_col': np.full(96, 0.125), 'dy': 0.125, 'mat_col': None, 'mat_el': None}
    assert fe.strip_plan(0, 16) is None                  # 6 columns per strip: narrower than any halo
    fe._grid['dx_col'] = np.concatenate((np.full(48, 0.125), np.full(48, 0.25)))
    assert fe.strip_plan(0, 2) is None                   # non-uniform elements: no matrix-free grid operator


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import scipy.sparse as sp
        from oracle import oracle as O
        from oracle.solve_ref import RefSolver
        fn = FE.host_transport(dist, rank, world)
        # --- all-reduces of the transport
        a = np.arange(5, dtype=float) * (rank + 1)
        fn(a, 0)
        assert np.allclose(a, np.arange(5) * sum(range(1, world + 1)))
        b = np.array([3.0 - rank, 7.0 + rank])
        fn(b, 3)
        assert np.allclose(b, [3.0 - (world - 1), 7.0])
        ii = np.array([rank == 1, 0, 1, 5], dtype=np.int32)
        fn(ii, 0)
        assert list(ii) == [1, 0, world, 5 * world]
        # --- strip geometry
        NX, NY = 32 * world, 8
        fe = make_model(NX, NY)
        plan = fe.strip_plan(rank, world, coarse_level=2)
        assert plan is not None and plan['W'] == 16
        W, c0, c1, g0, g1 = plan['W'], plan['c0'], plan['c1'], plan['g0'], plan['g1']
        nyn = NY + 1
        has_left, has_right = rank > 0, rank < world - 1
        # global reference vector (same seed everywhere); this rank knows it on the node columns c0..c1 only
        p_glob = np.random.default_rng(3).normal(size=(NX + 1, nyn, 2))
        p_loc = np.full((g1 - g0 + 1, nyn, 2), np.nan)
        p_loc[c0 - g0:c1 - g0 + 1] = p_glob[c0:c1 + 1]
        n = W * nyn * 2
        buf = np.zeros(2 * n)
        oc0, oc1 = c0 - g0, c1 - g0
        if has_left:
            buf[:n] = p_loc[oc0 + 1:oc0 + 1 + W].ravel()
        if has_right:
            buf[n:] = p_loc[oc1 - W:oc1].ravel()
        fn(buf, 100)
        if has_left:
            p_loc[oc0 - W:oc0] = buf[:n].reshape(W, nyn, 2)
        if has_right:
            p_loc[oc1 + 1:oc1 + 1 + W] = buf[n:].reshape(W, nyn, 2)
        assert not np.any(np.isnan(p_loc))
        assert np.array_equal(p_loc, p_glob[g0:g1 + 1])          # the halo carries the neighbours' values, bit for bit
        # --- operator of the local elements vs the global one
        ref = RefSolver(fe)
        ref.elstiff = np.array(ref.CVs[ref.mat_id])
        ref.elstiff[:, 0] *= 1. + 0.3 * np.sin(np.arange(ref.nel))   # element-wise varying tangents
        Kfull = ref.setupK()
        Kel = O.kel_batch(ref.lxy, ref.mat_id, ref.thick, ref.ps, ref.CVs, ref.Es, ref.nus, ref.elstiff)
        sel = slice(g0 * NY * 64, g1 * NY * 64)                   # elements of the local window (columns g0..g1)
        Kloc = sp.coo_matrix((Kel.ravel()[sel], (ref.rows[sel], ref.cols[sel])), shape=Kfull.shape).tocsr()
        pg = np.zeros(fe.Ndof)
        pg[2 * g0 * nyn:2 * (g1 + 1) * nyn] = p_loc.ravel()       # local vector at its global position, zero elsewhere
        q_loc = (Kloc @ pg).reshape(NX + 1, nyn, 2)
        q_glob = (Kfull @ p_glob.ravel()).reshape(NX + 1, nyn, 2)
        lo = g0 + (1 if has_left else 0)                          # outermost column of an interior edge: rows incomplete
        hi = g1 - (1 if has_right else 0)
        scale = np.max(np.abs(q_glob))
        assert np.max(np.abs(q_loc[lo:hi + 1] - q_glob[lo:hi + 1])) < 1e-12 * scale
        if has_left:
            assert np.max(np.abs(q_loc[g0] - q_glob[g0])) > 1e-6 * scale     # ... and they really are
        ok = True
    except Exception:  # pragma: no cover
        import traceback
        ok = traceback.format_exc()
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_halo_exchange_and_local_operator_gloo(world):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok in res:
        assert ok is True, (rank, ok)
