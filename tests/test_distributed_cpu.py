"""World-size-2 gloo tests (CPU) of the sharded path's host logic: x-strip ownership, the scalar
collectives the façade performs, and the identity the per-CG-step all-reduce relies on
(sum over strips of K_strip p == K p), checked with the CPU oracle's assembly."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import pylabfea_amd as FE


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def make_model(n=12):
    mat = FE.Material()
    mat.elasticity(E=200.e3, nu=0.3)
    mat.plasticity(sy=100., hill=[0.7, 1., 1.4, 1., 1.2, 0.8], khard=100., sdim=6)
    fe = FE.Model(dim=2)
    fe.geom([4.], LY=4.)
    fe.assign([mat])
    fe.bcleft(0.)
    fe.bcbot(0.)
    fe.bcright(0., 'force')
    fe.bctop(0.002 * fe.leny, 'disp')
    fe.mesh(NX=n, NY=n)
    return fe


def test_strip_partition():
    fe = make_model(13)
    for nranks in (1, 2, 3, 4, 8):
        edges = [fe.strip_range(r, nranks) for r in range(nranks)]
        assert edges[0][0] == 0 and edges[-1][1] == fe.Nel
        for (a0, a1), (b0, b1) in zip(edges[:-1], edges[1:]):
            assert a1 == b0 and a0 <= a1
        for a0, a1 in edges:
            assert a0 % fe._NY == 0 and a1 % fe._NY == 0          # whole element columns
        sizes = [a1 - a0 for a0, a1 in edges]
        assert max(sizes) - min(sizes) <= fe._NY                   # balanced to one column
        # owned node range is contiguous and neighbours share exactly one node column
        for (a0, a1), (b0, b1) in zip(edges[:-1], edges[1:]):
            if a1 > a0 and b1 > b0:
                na = set(fe._conn[a0:a1].ravel())
                nb = set(fe._conn[b0:b1].ravel())
                assert len(na & nb) == fe.NnodeY


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from oracle.solve_ref import RefSolver
        fe = make_model(12)
        fe._shard = (rank, world, None)             # host-side collectives only; no engine is created
        fe._host_allreduce = FE.host_transport(dist, rank, world)
        # --- scalar collectives of the façade
        ch, cv = fe._allreduce_flags(rank == 1, rank == 0)
        assert bool(ch) is True and bool(cv) is False
        cnt, mn, s = fe._allreduce_scf(10 + rank, 0.5 - 0.1 * rank, 3.0 * (rank + 1))
        assert cnt == 21 and abs(mn - 0.4) < 1e-15 and abs(s - 9.0) < 1e-12
        tot = fe._allreduce_sum(np.arange(18, dtype=float) * (rank + 1))
        assert np.allclose(tot, np.arange(18) * 3.)
        # --- sharded assembly + SpMV + all-reduce == global SpMV
        ref = RefSolver(fe)
        ref.elstiff = np.array(ref.CVs[ref.mat_id])
        rng = np.random.default_rng(5)
        ref.elstiff += 1e3 * rng.normal(size=(1, 36))[:, :] * 0  # keep symmetric elastic tangents
        Kfull = ref.setupK()
        e0, e1 = fe.strip_range(rank, world)
        Kel = O.kel_batch(ref.lxy, ref.mat_id, ref.thick, ref.ps, ref.CVs, ref.Es, ref.nus, ref.elstiff)
        import scipy.sparse as sp
        sel = slice(e0 * 64, e1 * 64)
        Kloc = sp.coo_matrix((Kel.ravel()[sel], (ref.rows[sel], ref.cols[sel])), shape=Kfull.shape).tocsr()
        p = np.random.default_rng(7).normal(size=fe.Ndof)
        qv = torch.from_numpy(Kloc @ p)
        dist.all_reduce(qv)
        assert np.max(np.abs(qv.numpy() - Kfull @ p)) < 1e-9 * np.max(np.abs(Kfull @ p))
        # the strip's rows are non-zero only inside its contiguous node range
        nodes = np.unique(fe._conn[e0:e1])
        nz = np.unique(Kloc.nonzero()[0] // 2)
        assert nz.min() >= nodes.min() and nz.max() <= nodes.max()
        assert np.array_equal(nodes, np.arange(nodes.min(), nodes.max() + 1))
        q_ok = True
    except Exception:  # pragma: no cover
        import traceback
        q_ok = traceback.format_exc()
    q.put((rank, q_ok))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok in res:
        assert ok is True, (rank, ok)
