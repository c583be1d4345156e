#!/usr/bin/env python3
"""Scipy prototype (decision aid, not product): PCG iteration counts when the V(2,2) cycle is multiplicative on levels < la
and ADDITIVE (all levels smoothed concurrently from the restricted right-hand side, corrections summed on the way up) from
level la down -- 2 dependent launches per level + 1 instead of 6.
    python tools/probes/mg_proto_additive.py gpurun_out/tang128.npz"""
import sys

import numpy as np

sys.argv, argv = sys.argv[:1], sys.argv[1:]
from mg_proto_coarsen4 import hierarchy, dirichlet_mask, assemble  # noqa: E402


def smooth0(L, b, om, nu):
    x = np.zeros_like(b)
    for _ in range(nu):
        x += om * L['dinv'] * (b - L['K'] @ x)
    return x


def additive(levels, l, b, om, nu, scale):
    L = levels[l]
    if 'lu' in L:
        return L['lu'].solve(b)
    return scale * smooth0(L, b, om, nu) + L['P'] @ additive(levels, l + 1, L['P'].T @ b, om, nu, scale)


def vcycle(levels, l, b, om, nu, la, scale, nua):
    L = levels[l]
    if 'lu' in L:
        return L['lu'].solve(b)
    if l >= la:
        return additive(levels, l, b, om, nua, scale)
    x = smooth0(L, b, om, nu)
    r = b - L['K'] @ x
    x += L['P'] @ vcycle(levels, l + 1, L['P'].T @ r, om, nu, la, scale, nua)
    for _ in range(nu):
        x += om * L['dinv'] * (b - L['K'] @ x)
    return x


def pcg(levels, b, la, scale=1., nua=2, om=0.65, nu=2, rtol=1e-10, maxit=400):
    K = levels[0]['K']
    x = np.zeros_like(b)
    r = b.copy()
    zv = vcycle(levels, 0, r, om, nu, la, scale, nua)
    p = zv.copy()
    rz = r @ zv
    bn = np.linalg.norm(b)
    for it in range(1, maxit + 1):
        q = K @ p
        a = rz / (p @ q)
        x += a * p
        r -= a * q
        if np.linalg.norm(r) <= rtol * bn:
            return it
        zv = vcycle(levels, 0, r, om, nu, la, scale, nua)
        rz2 = r @ zv
        p = zv + (rz2 / rz) * p
        rz = rz2
    return maxit


def run(name, n, D0):
    m0 = dirichlet_mask(n)
    Kf = assemble(n, D0)
    top = np.zeros((n + 1, n + 1, 2))
    top[:, n, 1] = 1.
    b = -(Kf @ top.ravel()) * m0
    b2 = np.random.default_rng(0).standard_normal(len(b)) * m0
    lv = hierarchy(n, D0, [2] * 12)
    print('== %s, n = %d, levels %s' % (name, n, [L['nx'] for L in lv]))
    print('  multiplicative (built)          its tension %3d random %3d' % (pcg(lv, b, 99), pcg(lv, b2, 99)))
    for la in (1, 2, 3):
        for scale in (1., 0.5):
            for nua in (2, 4):
                print('  additive from level %d scale %.1f nu %d its tension %3d random %3d' % (la, scale, nua, pcg(lv, b, la, scale, nua), pcg(lv, b2, la, scale, nua)))


for f in argv:
    z = np.load(f)
    n = int(z['n'])
    run('dumped tangent field ' + f, n, z['D'])
    E, nu = 200e3, 0.3
    lam, mu = E * nu / ((1 + nu) * (1 - 2 * nu)), E / (2 * (1 + nu))
    run('homogeneous elastic', n, np.tile(np.array([lam + 2 * mu, lam, 0., lam + 2 * mu, 0., mu]), (n * n, 1)))
