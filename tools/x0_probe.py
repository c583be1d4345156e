#!/usr/bin/env python3
"""Experiment: does extrapolating the PCG start vector save iterations?  In steady plastic flow every load step repeats the
pattern (solution A with the old stiffness, solution B after the tangent update); start the B-solve from
x0 = A + (B_prev - A_prev) instead of x0 = A.  Uses the Python driver (PLFX_NATIVE_STEP=0) and host round trips of du -
iteration counts only, timings are meaningless.  x0_probe.py [n] [steps] [mode]"""
import os
import sys
os.environ['PLFX_NATIVE_STEP'] = '0'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import pylabfea_amd as FE
from pylabfea_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
mode = sys.argv[3] if len(sys.argv) > 3 else 'extrap'

hist = {'delta': None, 'changed': False, 'log': []}
orig = FE.Model._solve_lin
orig_sweep = _lib.Context.sweep


def sweep(self, nit):
    ch, cv = orig_sweep(self, nit)
    hist['changed'] = hist['changed'] or ch
    return ch, cv


_lib.Context.sweep = sweep


def patched(self, eng, bc, warm):
    upd = hist['changed'] and warm          # the stiffness differs from the one of the previous solve
    x_before = eng.state_get(_lib.ST_DU).copy() if upd else None
    if upd and mode == 'extrap' and hist['delta'] is not None:
        eng.state_set(_lib.ST_DU, x_before + float(sys.argv[4] if len(sys.argv) > 4 else 1.) * hist['delta'])
    orig(self, eng, bc, warm)
    if upd:
        hist['delta'] = eng.state_get(_lib.ST_DU) - x_before   # what this tangent update did to the solution
        hist['changed'] = False
    hist['log'].append(self.solver_stats[-1][0])


FE.Model._solve_lin = patched
fe = bench.tension_model(FE, bench.hill_material(FE), n, 0.005, device=0)
fe._max_load_steps = steps
fe.solve(min_step=50)
print('mode %s n=%d: PCG its per solve %s  total %d' % (mode, n, hist['log'], sum(hist['log'])))
print('sgl_yy %.9f' % fe.sgl[-1][1])
