import sys, warnings, numpy as np
sys.path.insert(0, '/root/repo')
import pylabfea_amd as FE
def run(nx, ny, operator):
    m = FE.Material(); m.elasticity(E=200.e3, nu=0.3)
    m.plasticity(sy=100., hill=[0.7, 1., 1.4, 1., 1.2, 0.8], khard=100., sdim=6)
    fe = FE.Model(dim=2, planestress=False); fe.operator = operator
    fe.geom([4.], LY=4. * ny / nx); fe.assign([m]); fe.bcleft(0.); fe.bcbot(0.); fe.bcright(0., 'force'); fe.bctop(0.004 * fe.leny, 'disp')
    fe.mesh(NX=nx, NY=ny); fe._max_load_steps = 10
    with warnings.catch_warnings():
        warnings.simplefilter('ignore'); fe.solve(min_step=12)
    e = fe._engine
    print(nx, ny, 'operator', e.operator_info(), 'precond', e.precond_info(), 'its', sum(q[0] for q in fe.solver_stats), 'sgl_yy %.9f' % fe.sgl[-1][1], flush=True)
    return np.array(fe.u), fe._state('sig').copy()
for nx, ny in ((255, 257), (300, 201), (77, 51)):
    a = run(nx, ny, None); b = run(nx, ny, 0)
    print('   u rel diff %.2e  sig rel diff %.2e' % (np.max(np.abs(a[0] - b[0])) / np.max(np.abs(a[0])), np.max(np.abs(a[1] - b[1])) / np.max(np.abs(a[1]))))
