"""Non-proportional laminates (model.py:826-847: dx = LS[i] / nes[i] per section): multigrid V-cycle of the uniform grid as the
preconditioner of the exact operator (round 5) against Jacobi-PCG (what such meshes ran before) -- config-3 style
workload, two plastic materials in five sections.  python tools/probes/laminate_probe.py "((nx, ny, (sections)), ...)" """
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pylabfea_amd as FE


def run(nx, ny, LS, precond=None, operator=None, steps=12, same_mat=False):
    ma = FE.Material(num=1)
    ma.elasticity(E=200.e3, nu=0.3)
    ma.plasticity(sy=150., khard=500., sdim=6)
    mb = FE.Material(num=2)
    mb.elasticity(E=120.e3, nu=0.33)
    mb.plasticity(sy=90., hill=[0.8, 1.1, 1.3, 1., 0.9, 1.2], khard=300., sdim=6)
    fe = FE.Model(dim=2, planestress=True)
    fe.geom(list(LS), LY=float(sum(LS)) * ny / nx)
    fe.assign([ma if (same_mat or i % 2 == 0) else mb for i in range(len(LS))])
    fe.bcleft(0.); fe.bcbot(0.); fe.bcright(0., 'force'); fe.bctop(0.004 * fe.leny, 'disp')
    fe.mesh(NX=nx, NY=ny)
    fe.precond, fe.operator = precond, operator
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        eng = fe._ensure_engine()
        marks = {}

        def hook(il):
            if il == 6: eng.sync(); marks['t0'] = time.perf_counter(); marks['s0'] = len(fe.solver_stats)
            if il == steps: eng.sync(); marks['t1'] = time.perf_counter(); marks['s1'] = len(fe.solver_stats)
        fe._step_hook = hook; fe._max_load_steps = steps
        fe.solve(min_step=20)
    its = [q[0] for q in fe.solver_stats[marks['s0']:marks['s1']]]
    pi = eng.precond_info()
    dx = np.unique(np.round(fe._grid['dx_col'], 12)) if hasattr(fe, '_grid') else []
    print('%5d x %5d LS %s: widths max/min %.4f  preconditioner %-9s (%2d levels) operator %-11s %9.3f ms per load step (6..%d)  %d solves, %6d iterations (max %5d)  sgl_yy %.7f'
          % (nx, ny, LS, (max(dx) / min(dx)) if len(dx) else 0., 'multigrid' if pi[0] == 1 else 'Jacobi', pi[1], 'matrix-free' if eng.operator_info()[0] else 'block-ELL',
             1e3 * (marks['t1'] - marks['t0']) / (steps - 6), steps, len(its), sum(its), max(its), fe.sgl[-1][1]), flush=True)
    out = (np.array(fe.u), fe._state('sig').copy(), list(fe.niter))
    fe._drop_engine()
    return out


if __name__ == '__main__':
    import ast
    cases = ast.literal_eval(sys.argv[1]) if len(sys.argv) > 1 else ((52, 8, (2, 1, 2, 1, 2)), (256, 256, (3, 1, 2, 1, 2)), (512, 512, (3, 1, 2, 1, 2)))
    for nx, ny, LS in cases:
        a = run(nx, ny, LS)
        b = run(nx, ny, LS, operator=0) if nx * ny <= 600000 else None      # block-ELL Krylov operator (exact class shapes), same V-cycle
        j = run(nx, ny, LS, precond=0) if nx * ny <= 300000 else None       # Jacobi-PCG: what these meshes ran before
        for tag, o in (('block-ELL', b), ('Jacobi', j)):
            if o is not None:
                print('      vs %-9s: u %.2e  sig %.2e (relative), K-iterations equal: %s'
                      % (tag, np.max(np.abs(a[0] - o[0])) / np.max(np.abs(o[0])), np.max(np.abs(a[1] - o[1])) / np.max(np.abs(o[1])), a[2] == o[2]), flush=True)
