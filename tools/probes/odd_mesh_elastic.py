"""cold elastic solves on even / odd meshes: PCG iterations of the one solve (compare tools/probes/mg_proto_oddghost.py)"""
import os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pylabfea_amd as FE
for nx, ny in ((32, 32), (32, 31), (31, 32), (31, 31), (64, 63), (128, 127), (127, 128), (128, 128), (512, 511), (511, 512)):
    m = FE.Material(); m.elasticity(E=200.e3, nu=0.3)
    fe = FE.Model(dim=2, planestress=False)
    fe.geom([4.], LY=4. * ny / nx); fe.assign([m]); fe.bcleft(0.); fe.bcbot(0.); fe.bcright(0., 'force'); fe.bctop(0.001 * fe.leny, 'disp')
    fe.mesh(NX=nx, NY=ny)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fe.solve()
    pi = fe._engine.precond_info()
    print('%4d x %4d  %s %d levels  PCG iterations %s  sig_yy %.9f' % (nx, ny, 'MG' if pi[0] == 1 else 'Jacobi', pi[1], [q[0] for q in fe.solver_stats], fe.sgl[-1][1]))
