"""``Material`` façade: the reference's constitutive API for the hot path, evaluated by libplfx on
the MI355X.

Mirrors pylabfea.Material (reference /root/reference/src/pylabfea/material.py) for the methods on
the path named by BASELINE.json: ``elasticity`` (:2401), ``plasticity`` (:2466), ``response`` (:207),
``calc_yf`` (:348), ``ML_full_yf`` (:414), ``calc_seq`` (:576), ``calc_fgrad`` (:704), ``get_sflow``
(:974), ``epl_dot`` (:1009), ``C_tan`` (:1057) and the test harness ``calc_properties`` (:3062).
Same names, argument meaning and error behaviour; arguments are never mutated.

Out of scope here: SVC *training*, data import, plotting, texture/work-hardening
features.  A trained SVC enters through :meth:`Material.set_svc` / :meth:`Material.from_sklearn`.
sdim=3 flow rules use the reference's axis-tracking principal stresses (exact for plane states);
Tresca and Barlat Yld2004-18p are equivalent stresses only (the reference has no normal for them).
ML materials: 6 stress features (sdim=6) or the 2 features (seq, polar angle) of ``setup_yf_SVM_3D`` (sdim=3).
"""
import hashlib
import os
import warnings

import numpy as np

from . import _lib
from .basic import eps_eq, sig_dev, sig_eq_j2, sig_polar_ang, sig_princ, yf_tolerance

_point_ctx = {}        # shared contexts for point evaluations, one per GPU (each remembers the record it holds)


def point_device():
    """GPU of the point-evaluation context: PLFX_DEVICE (an explicit choice: must name a visible GPU, like
    ``Model(device=...)``), else LOCAL_RANK (one process per GPU under torchrun; taken modulo the number of visible GPUs so
    that several ranks sharing one GPU -- host transport, tests -- all find a device), else 0."""
    n = _lib.device_count()
    v = os.environ.get('PLFX_DEVICE')
    if v is not None and v.strip().isdigit():
        if n > 0 and int(v) >= n:
            raise ValueError('PLFX_DEVICE=%s, but only %d GPU(s) are visible' % (v.strip(), n))
        return int(v)
    v = os.environ.get('LOCAL_RANK')
    if v is not None and v.strip().isdigit():
        return int(v) % n if n > 0 else 0
    return 0


try:    # the support-vector tables (~90 KB) are digested on every point call (in-place edits must be seen): xxh3 does it in
    import xxhash as _xxhash   # ~8 us where blake2b needs ~100 us; blake2b when the module is absent
except ImportError:  # pragma: no cover
    _xxhash = None


def _table_digest(a):
    """16-byte digest of the CONTENT of a contiguous array"""
    if _xxhash is not None:
        return _xxhash.xxh3_128_digest(a.data)
    return hashlib.blake2b(a.data, digest_size=16).digest()


def _ctx():
    dev = point_device()
    if dev not in _point_ctx:
        _point_ctx[dev] = _lib.Context(dev)
    return _point_ctx[dev]


def close_point_contexts():
    """release the shared point-evaluation contexts (HBM of the support-vector tables, streams)"""
    for c in _point_ctx.values():
        c.close()
    _point_ctx.clear()


class Material(object):
    """Material card: elastic constants, plastic parameters, optional trained SVC yield function.

    Attributes follow the reference (``sy, khard, hill, drucker, E, nu, C11, C12, C44, CV, sdim,
    ML_yf, dev_only, scale_seq, gam_yf, msg, prop, propJ2, sigeps`` ...).
    """

    def __init__(self, name='Material', num=1):
        self.name = name
        self.num = num
        self.E = self.nu = self.CV = self.C11 = self.C12 = self.C44 = None
        self.sy = None          # elasticity only unless sy is set (material.py:159)
        self.sy0 = None
        self.khard = None
        self.drucker = None
        self.lhs = None
        self.hill = None
        self.hill_3p = False
        self.hill_6p = False
        self.tresca = False
        self.barlat = False
        self.sdim = None
        self.ML_yf = False
        self.ML_grad = False
        self.dev_only = False
        self.scale_seq = None
        self.gam_yf = None
        self.C_yf = None
        self.Ndof = 2
        self.svc = None         # dict(sv, dual, intercept) of the trained SVC
        self.msg = {'yield_fct': None, 'gradient': None, 'nsteps': 0, 'equiv': None}
        lcs = ('stx', 'sty', 'et2', 'ect')
        self.prop = {k: {'ys': None, 'seq': None, 'eeq': None, 'peeq': None, 'style': None, 'name': None}
                     for k in lcs}
        self.propJ2 = {k: {'ys': None, 'seq': None, 'eeq': None, 'peeq': None} for k in lcs}
        self.sigeps = {k: {'sig': None, 'eps': None, 'epl': None} for k in lcs}
        self._version = 0

    # ------------------------------------------------------------------ definition
    def elasticity(self, C11=None, C12=None, C44=None, CV=None, E=None, nu=None):
        """Define elastic properties (material.py:2401-2464): one of (E, nu), (C11, C12, C44) or the full 6 x 6 matrix;
        the other descriptions are derived from the one given.  Same exceptions and messages as the reference."""
        clash = 'Error: Inconsistent definition of material parameters: '
        cubic = (C11, C12, C44)
        if E is not None:      # isotropic
            if nu is None:
                raise ValueError(clash + 'Only E provided')
            if any(c is not None for c in cubic):
                raise ValueError(clash + 'E provided together with C_ij')
            hh = E / ((1. + nu) * (1. - 2. * nu))
            cubic = ((1. - nu) * hh, nu * hh, (0.5 - nu) * hh)
        elif C11 is not None:  # cubic
            if nu is not None:
                raise ValueError(clash + 'nu provided together with C_ij')
            if None in (C12, C44):
                raise ValueError(clash + 'C_12 or C_44 values missing')
        elif CV is not None:   # general: Poisson ratio and modulus of the [100] direction
            self.CV = np.array(CV, dtype=float)
            cubic = (self.CV[0, 0], self.CV[0, 1], self.CV[3, 3])
        else:
            raise ValueError('elasticity: Inconsistent definition of material parameters')
        self.C11, self.C12, self.C44 = cubic
        if E is None:
            nu = self.C12 / (self.C11 + self.C12)
            E = 2 * self.C44 * (1 + nu)
        self.E, self.nu = E, nu
        if CV is None:
            M = np.full((6, 6), 0.)
            M[:3, :3] = self.C12
            for k in range(3):
                M[k, k], M[k + 3, k + 3] = self.C11, self.C44
            self.CV = M
        self._version += 1

    def plasticity(self, sy=None, sdim=6, drucker=0., khard=0., tresca=False, barlat=None,
                   barlat_exp=None, hill=None, hill_3p=None, hill_6p=None, rv=None, lhs=None):
        """Define plastic parameters (material.py:2466-2594).  Same exceptions, warnings and messages as the reference;
        the Hill coefficients end up as 6 numbers for sdim = 6 and 3 for sdim = 3, with `hill_6p` / `hill_3p` telling
        which equivalent-stress form applies (both False: J2)."""
        if sy < 0.:
            raise ValueError('Initial yield strength cannot be negative.')
        if khard < 0.:
            warnings.warn('Strain softening not supported. khard is set to 0.')
        if lhs is not None:
            # the reference stores lhs but calc_seq evaluates `if self.lhs:` on the array
            # (material.py:642), which raises for any 3-vector: the option cannot run there either
            raise NotImplementedError('lhs (Liu-Huang-Stout asymmetry) is not supported')
        if sdim not in (3, 6):
            raise ValueError('{} in plasticity: sdim must be either 3 or 6'.format(self.name))
        if barlat is not None and len(barlat) != 18:
            raise ValueError('plasticity: barlat must hold the 18 Yld2004-18p coefficients')
        if self.sdim not in (None, sdim):
            print('plasticity: Parameter sdim is changed. New value:', sdim)
        self.sdim, self.sy0, self.sy = sdim, sy, sy
        self.khard, self.drucker, self.lhs = max(khard, 0.), drucker, None
        if hill is not None:
            if rv is not None:
                warnings.warn('plasticity: Both, hill and rv, have been provided. Using Hill parameters.')
            hill = list(hill)
        elif rv is None:
            hill = [1.] * sdim
        else:  # yield-stress ratios -> Hill coefficients
            if len(rv) != sdim:
                raise ValueError(f'plasticity: wrong dimension of yield stress ratios, must be {sdim}')
            q = (1. / np.array(rv, dtype=float)) ** 2
            hill = list(q[:3] + np.roll(q[:3], -1) - np.roll(q[:3], -2)) + list(q[3:])
        given = len(hill)
        if hill_6p is None and hill_3p is None:  # decided by the number of coefficients; three ones are J2, not Hill
            six = given == 6
            three = not six and not (hill[0] == 1. and hill[1] == 1. and hill[2] == 1.)
        else:
            six, three = hill_6p, hill_3p
        if six and given != 6:
            raise ValueError('plasticity: When hill_6p is set True, 6 Hill parameters must be provided')
        if three and given != 3:
            raise ValueError('plasticity: When hill_3p is set True, only 3 Hill parameters can be provided')
        if six and sdim == 3:
            warnings.warn('plasticity: 6 Hill parameters are provided, but sdim=3; ignoring shear parameters')
            six, three = False, True
            del hill[3:]
        if three and sdim == 6:
            print('Material', self.name)
            warnings.warn('plasticity: 3 Hill parameters are provided, but sdim=6; shear parameters set to 1')
            six, three = True, False
        if sdim == 6 and len(hill) == 3:
            hill += [1., 1., 1.]
        self.hill_6p, self.hill_3p = bool(six), bool(three)
        self.hill = np.array(hill, dtype=float)
        self.tresca = bool(tresca)
        self.barlat = barlat is not None
        if self.barlat:
            self.barlat_par = np.array(barlat, dtype=float)
            self.barlat_exp = barlat_exp
            b = self.barlat_par  # the two linear maps of the stress deviator (material.py:2578-2591)
            for name, c in (('Bar_m1', b[0:9]), ('Bar_m2', b[9:18])):
                m = np.zeros((6, 6))
                m[0, 1], m[0, 2], m[1, 0], m[1, 2], m[2, 0], m[2, 1] = -c[0], -c[1], -c[2], -c[3], -c[4], -c[5]
                m[3, 3], m[4, 4], m[5, 5] = c[6], c[7], c[8]
                setattr(self, name, m)
        self._version += 1

    def set_svc(self, support_vectors, dual_coef, intercept, gamma, scale_seq, dev_only=False, C=None, scale_wh=None):
        """Install a trained RBF-SVC yield function (what ``train_SVC`` leaves in ``svm_yf``,
        ``gam_yf`` and ``scale_seq``; material.py:398-405, 797-807).  ``dual_coef``/``intercept`` are
        scikit-learn's public ``dual_coef_[0]`` / ``intercept_[0]``."""
        if self.sy is None:
            raise ValueError('set_svc: call elasticity() and plasticity(sy=..., sdim=6) first')
        nfeat = 6 if self.sdim == 6 else 2   # sdim=3: (seq_J2/scale - 1, polar angle/pi), material.py:2331-2333
        sv = np.ascontiguousarray(support_vectors, dtype=float)
        # work-hardening-aware SVC (train_SVC on Data(wh_data=True), material.py:2342-2346): 6 stress features, the plastic
        # strain / scale_wh (6), accumulated strain, max. stress / scale_seq, flag  ->  Ndof = 15, ind_wh = 6
        self.whdat = bool(self.sdim == 6 and sv.ndim == 2 and sv.shape[1] == 15)
        if self.whdat:
            if scale_wh is None or not scale_wh > 0.:
                raise ValueError('set_svc: 15-feature support vectors (work-hardening data) need scale_wh > 0')
            nfeat = 15
            self.ind_wh = 6
            self.scale_wh = float(scale_wh)
        if sv.ndim != 2 or sv.shape[1] != nfeat:
            raise ValueError('set_svc: support vectors must have shape (nsv, %d) for sdim=%d' % (nfeat, self.sdim))
        dual = np.ascontiguousarray(dual_coef, dtype=float).reshape(-1)
        if len(dual) != len(sv):
            raise ValueError('set_svc: dual_coef and support_vectors differ in length')
        self.svc = dict(sv=sv, dual=dual, intercept=float(intercept))
        self.gam_yf = float(gamma)
        self.C_yf = C
        self.scale_seq = float(scale_seq)
        self.dev_only = bool(dev_only)
        self.ML_yf = True
        self.Ndof = nfeat
        if self.khard and not self.whdat:
            # calc_fgrad of an ML material resets self.khard to 0 on every call (material.py:812-814)
            warnings.warn('set_svc: khard of an ML material is reset to 0 by the reference flow rule')
            self.khard = 0.
        self._version += 1

    @classmethod
    def from_sklearn(cls, svm, scale_seq, E, nu, sy, name='ML-material', dev_only=False):
        """Build an ML material from a fitted ``sklearn.svm.SVC`` (RBF kernel, 6 stress features)."""
        m = cls(name=name)
        m.elasticity(E=E, nu=nu)
        m.plasticity(sy=sy, sdim=6)
        gamma = svm._gamma if hasattr(svm, '_gamma') else svm.gamma
        m.set_svc(svm.support_vectors_, svm.dual_coef_[0, :], svm.intercept_[0], gamma, scale_seq,
                  dev_only=dev_only, C=getattr(svm, 'C', None))
        return m

    # ------------------------------------------------------------------ SVC parameter files (reference wire format)
    def export_MLparam(self, sname, source=None, file=None, path='../../models/', descr=None, param=None):
        """Write the trained SVC in the reference's Abaqus-UMAT layout (material.py:2130-2271): ``file-svm.csv``
        with 8 numbers per line -- header slots 0..28 (nsv, Ndof, C11, C12, C44, intercept, gamma, epc, scale_seq,
        scale_wh, C22, C33, C13, C23, C55, C66, dev_only flag, Nset, scale_text), dual coefficients from slot 29,
        then the support vectors row by row -- and ``file-svm_meta.json``."""
        import datetime, getpass, json, platform
        if not self.ML_yf:
            raise AttributeError('export_MLparam: No ML flow rule defined.')
        if descr is not None and param is not None and len(descr) != len(param):
            raise ValueError('Lists for descr and param must have the same lengths.')
        file = (path if path.endswith('/') else path + '/') + ('abq_' + self.name if file is None else file)
        dc, sv = self.svc['dual'], self.svc['sv']
        nsv, ndof = sv.shape
        nlin = int((nsv * (ndof + 1) + 30) / 8) + 1
        ndata = nlin * 8
        head = np.zeros(29)
        head[:10] = (nsv, ndof, self.C11, self.C12, self.C44, self.svc['intercept'], self.gam_yf,
                     float(getattr(self, 'epc', 0.) or 0.), self.scale_seq,
                     self.scale_wh if getattr(self, 'whdat', False) else 1.)
        head[10:16] = -1 if self.CV is None else [self.CV[i, j] for i, j in ((1, 1), (2, 2), (0, 2), (1, 2), (4, 4), (5, 5))]
        head[16:19] = (-1. if self.dev_only else 0., 1, 1.)   # dev_only flag, Nset, scale_text
        body = np.concatenate([head, dc, sv.ravel()])
        props = np.concatenate([body, np.zeros(ndata - len(body))])
        np.savetxt(file + '-svm.csv', props.reshape((nlin, 8)), delimiter=', ', newline='\n')
        descr = ([] if descr is None else list(descr)) + ['Ndata', 'gamma', 'C']
        param = ([] if param is None else list(param)) + [ndata, self.gam_yf, self.C_yf]
        sys_info = platform.uname()
        from . import __version__ as vers
        meta = {
            'Info': {'Owner': getpass.getuser(), 'Institution': None, 'Date': str(datetime.date.today()),
                     'Description': 'SVC-parameters for plasticity model', 'Method': 'Support Vector Classification',
                     'System': {'sysname': sys_info[0], 'nodename': sys_info[1], 'release': sys_info[2],
                                'version': sys_info[3], 'machine': sys_info[4]}},
            'Model': {'Creator': 'pylabfea_amd', 'Version': vers, 'Repository': None, 'Input': source, 'Script': sname,
                      'Names': descr, 'Parameters': param},
            'Data': {'Class': 'SVC_parameters', 'Type': 'CSV', 'File': file + '-svm.csv', 'Separator': ',',
                     'Header': None, 'Format': (nlin, 8),
                     'Names': ['nsv', 'nsd', 'C11', 'C12', 'C44', 'rho', 'gamma', 'epc', 'scale_seq', 'scale_wh',
                               'C22', 'C33', 'C13', 'C23', 'C55', 'C66', 'Nset', 'scale_text[0:Nset]',
                               'dual_coef[0:nsv]', 'sup_vec[0:nsv,0:nsd]'],
                     'Units': {'Stress': 'MPa', 'Strain': 'None', 'Disp': 'mm', 'Force': 'N'}}}
        with open(file + '-svm_meta.json', 'w') as fp:
            json.dump(meta, fp, indent=2)

    def from_MLparam(self, name, path='../../models/'):
        """Define the material from a parameter file written by ``export_MLparam`` (of stock pyLabFEA or of this
        package): elastic constants, yield strength (= the stress scaling factor, ``scale_seq = sy`` at training,
        material.py:1161) and the SVC yield function.  The reference declares this method but raises
        ``ModuleNotFoundError`` (material.py:2688-2703); the layout read here is the one its ``export_MLparam``
        writes and its Abaqus UMAT reads (examples/UMAT/ml_umat.f:129-151), including the rule
        ``dev_only = props[16] < 0``, and the older layout of the files shipped under examples/UMAT/models (see below).
        Files with work-hardening or texture features (Ndof > 6) are refused."""
        import json
        if path and path[-1] != '/':
            path += '/'
        trunk = path + name
        for suffix in ('-svm.csv', '.csv', ''):
            if trunk.endswith(suffix) and suffix:
                trunk = trunk[:-len(suffix)]
                break
        props = np.loadtxt(trunk + '-svm.csv', delimiter=',').ravel()
        nsv, ndof = int(round(props[0])), int(round(props[1]))
        if nsv < 1 or 29 + nsv * (ndof + 1) > len(props):
            raise ValueError('from_MLparam: inconsistent header (nsv={}, Ndof={}, {} numbers)'.format(nsv, ndof, len(props)))
        if ndof not in (2, 6, 15):
            raise NotImplementedError('from_MLparam: {} features (texture descriptors) are not supported; only the 6 '
                                      'stress features, the 2 features of sdim=3, or 15 = 6 stress + 9 work-hardening '
                                      'features'.format(ndof))
        C11, C12, C44 = props[2], props[3], props[4]
        if np.all(props[10:16] == -1.):
            self.elasticity(C11=C11, C12=C12, C44=C44)
        else:
            CV = np.zeros((6, 6))
            CV[0, 0], CV[1, 1], CV[2, 2] = C11, props[10], props[11]
            CV[0, 1] = CV[1, 0] = C12
            CV[0, 2] = CV[2, 0] = props[12]
            CV[1, 2] = CV[2, 1] = props[13]
            CV[3, 3], CV[4, 4], CV[5, 5] = C44, props[14], props[15]
            self.elasticity(CV=CV)
        scale_seq = float(props[8])
        self.plasticity(sy=scale_seq, sdim=6 if ndof in (6, 15) else 3)
        C = None
        try:
            with open(trunk + '-svm_meta.json') as fp:
                meta = json.load(fp)
            names, par = meta['Model']['Names'], meta['Model']['Parameters']
            if 'C' in names:
                C = par[names.index('C')]
        except (IOError, OSError, KeyError, ValueError):
            pass
        dual = props[29:29 + nsv]
        sv = props[29 + nsv:29 + nsv * (ndof + 1)].reshape(nsv, ndof)
        # Layouts.  Current (export_MLparam of v4.4, material.py:2212-2214): slot 16 = -1 / 0 deviatoric-feature flag, slot 17 =
        # Nset, 18.. = scale_text.  Files shipped with the reference under examples/UMAT/models (written by pyLabFEA 4.3,
        # "v4.0 layout"): slot 16 = Nset (>= 1), 17.. = scale_text, no flag -- those versions trained the 6-feature SVC on
        # DEVIATORIC stresses (every support vector is trace-free; with full-stress features the J2 file would yield under
        # hydrostatic load), so the flag is recovered from the support vectors.  Dual coefficients start at slot 29 in both.
        self.mlparam_layout = 'v4.0' if props[16] >= 1. else 'v4.4'
        if self.mlparam_layout == 'v4.0':
            dev_only = bool(ndof == 6 and np.max(np.abs(sv[:, 0:3].sum(axis=1))) < 1e-9 * max(np.max(np.abs(sv)), 1e-300))
        else:
            dev_only = bool(props[16] < 0.)
        self.set_svc(sv, dual, props[5], props[6], scale_seq, dev_only=dev_only, C=C,
                     scale_wh=float(props[9]) if ndof == 15 else None)
        if ndof == 15:
            self.epc = float(props[7])
        return self

    # ------------------------------------------------------------------ records for libplfx
    def _record(self, CV, ana=False):
        """plfx_material record of this material with the element matrix CV."""
        if self.sy is None:
            return _lib.pack_material(_lib.ELASTIC, CV, E=self.E, nu=self.nu)
        if self.ML_yf and not ana:
            svc = dict(sv=self.svc['sv'], dual=self.svc['dual'], intercept=self.svc['intercept'],
                       gamma=self.gam_yf, scale_seq=self.scale_seq, dev_only=self.dev_only,
                       scale_wh=getattr(self, 'scale_wh', None))
            kind = _lib.SVC_WH if getattr(self, 'whdat', False) else (_lib.SVC6 if self.sdim == 6 else _lib.SVC3)
            return _lib.pack_material(kind, CV, E=self.E, nu=self.nu, sy=self.sy, khard=self.khard,
                                      hill=self.hill, drucker=self.drucker, svc=svc)
        kind = _lib.HILL6 if self.sdim == 6 else _lib.PRINC3
        if self.tresca:
            kind = _lib.TRESCA
        elif self.barlat:
            kind = _lib.BARLAT
        return _lib.pack_material(kind, CV, E=self.E, nu=self.nu, sy=self.sy, khard=self.khard,
                                  hill=self.hill, drucker=self.drucker,
                                  barlat=self.barlat_par if self.barlat else None,
                                  barlat_exp=self.barlat_exp if self.barlat else 0.,
                                  barlat_normal=bool(self.barlat and getattr(self, 'barlat_normal', False)))

    def enable_barlat_normal(self, on=True):
        """EXTENSION (not in the reference, which has no flow rule for Barlat materials: calc_fgrad raises,
        material.py:822-825): use the analytic normal of Yld2004-18p -- d seq / d sigma through the eigen-decompositions
        of the two linearly transformed deviators -- so that ``calc_fgrad``, ``response``, ``calc_properties`` and
        ``Model.solve`` work for a Barlat material with an associated flow rule, like they do for Hill materials."""
        if not self.barlat:
            raise AttributeError('enable_barlat_normal: material has no Barlat parameters')
        self.barlat_normal = bool(on)
        self._version += 1
        return self

    def _no_flow_rule(self):
        if self.barlat and not getattr(self, 'barlat_normal', False):
            raise ValueError('calc_fgrad: analytical gradient for Barlat not implemented')
        if self.tresca:
            raise ValueError('calc_fgrad: analytical gradient for Tresca not implemented')

    def _content_key(self, CV=None, ana=False, rec=None, parameters_only=False):
        """Digest of everything the device evaluates for this material: the packed ``plfx_material`` record (kind, elastic
        and plastic parameters, Hill / Barlat coefficients, SVC scalars) and the support-vector / dual-coefficient tables.
        Two materials share a key only if the GPU would compute the same numbers for them -- never because one was
        garbage-collected and the next landed at the same address, and an attribute edited in place (``m.hill[0] = ...``,
        which the reference honours on the next call, material.py:139-205) changes the key.  ``parameters_only`` leaves
        out what is STATE rather than a parameter: the hardening modulus of a work-hardening SVC material, which every
        gradient evaluation overwrites (material.py:808-814)."""
        if rec is None:
            rec = self._record(np.asarray(self.CV if CV is None else CV, dtype=float), ana=ana)
        m, keep = rec
        h = hashlib.blake2b(digest_size=16)
        sv_ptr, dual_ptr = m.sv, m.dual
        kh = m.khard
        m.sv = m.dual = None             # host addresses of the tables are not content
        if parameters_only and m.kind == _lib.SVC_WH:
            m.khard = 0.
        h.update(bytes(m))
        m.sv, m.dual, m.khard = sv_ptr, dual_ptr, kh
        for a in keep:
            h.update(_table_digest(np.ascontiguousarray(a)))
        return h.digest()

    def _load(self, CV=None, ana=False):
        """Make this material (with element matrix CV) material 0 of the shared point context.  The context keeps the last
        record it was given; it is re-sent whenever its CONTENT differs (see ``_content_key``)."""
        cv = np.asarray(self.CV if CV is None else CV, dtype=float)
        if cv.shape != (6, 6):
            raise ValueError('CV must be a (6,6) array')
        ctx = _ctx()
        rec = self._record(cv, ana=ana)
        key = self._content_key(rec=rec)
        if getattr(ctx, '_point_key', None) != key:
            ctx._point_key = None          # a failing set_materials must not leave a stale key behind
            ctx.set_materials([rec])
            ctx._point_key = key
        return ctx

    @staticmethod
    def _voigt(sig, name):
        """(3,),(6,),(N,3),(N,6) -> (N,6) Voigt (principal stresses padded with zero shear)."""
        s = np.asarray(sig, dtype=float)
        sh = s.shape
        single = sh in ((3,), (6,))
        if single:
            s = s[None, :]
        if s.ndim != 2 or s.shape[1] not in (3, 6):
            raise TypeError('Unknown format of stress in %s' % name)
        if s.shape[1] == 3:
            s = np.concatenate((s, np.zeros((len(s), 3))), axis=1)
        return np.ascontiguousarray(s), single

    def _princ_rows(self, s):
        """Principal-stress materials (sdim = 3: 3-parameter Hill; the 2-feature SVC, whose polar-angle feature is taken from
        the principal stresses, basic.py:68-104) see a Voigt stress through ``basic.sig_princ`` -- the
        general eigen-solver ``np.linalg.eig`` plus the axis-tracking re-ordering (basic.py:153-175) -- and the ORDER of the
        principal stresses enters the Hill form (material.py:667-670).  For plane states (every state of a 2-d model) the
        device reproduces that order in closed form; for states with out-of-plane shear it depends on LAPACK's eigenvalue
        order, so those rows are reduced HERE with the very same LAPACK call and handed to the device as diagonal states
        (whose order the device keeps): ``calc_seq`` and ``calc_yf`` then equal the reference for every stress state
        (fixture ``tests/golden/princ_general.npz``); ``ML_full_yf`` searches along the ray of the reduced state (the order
        does not change along a ray) and ``calc_fgrad`` with a (6,) stress takes its equivalent stress from ``calc_seq``
        (round 5; same fixture).  SCOPE: ``response`` is NOT covered -- it re-orders the principal stresses of a new stress
        in each of its up to 50 sub-steps (material.py:250-340), which the device does by the natural rule "axis i -> the
        eigenvector with the largest component i"; the reference's order there is whatever LAPACK's ``dgeev`` returns
        (measured on the fixture's 160 states: 94 % of the elastic and 25 % of the plastic results coincide).  Plane states
        -- all a 2-d ``Model`` ever produces -- are exact; ``response`` warns when it is handed anything else."""
        if self.sdim != 3 or self.tresca or self.barlat:   # (Tresca, Barlat: symmetric in the principal values)
            return s
        gen = (s[:, 3] != 0.) | (s[:, 4] != 0.)
        if not np.any(gen):
            return s
        s = s.copy()
        sp, _ = sig_princ(s[gen])
        s[gen, 0:3] = sp.reshape(-1, 3)
        s[gen, 3:6] = 0.
        return s

    # ------------------------------------------------------------------ constitutive functions
    def calc_seq(self, sig):
        """Generalised (Hill/Drucker) equivalent stress (material.py:576-676)."""
        s, single = self._voigt(sig, 'calc_seq')
        if self.sy is None:
            seq = sig_eq_j2(s)  # elastic material: J2 (material.py:637-640)
        else:
            seq = self._load(ana=True).seq(0, self._princ_rows(s))
            self.msg['equiv'] = ('6-parameter Hill, full Voigt stress'
                                 if self.sdim == 6 and not (self.tresca or self.barlat) else '3-parameter Hill')
        return seq[0] if single else seq

    def calc_seqB(self, sv):
        """Yld2004-18p (Barlat et al.) equivalent stress of Voigt stresses (material.py:678-702), evaluated by the device
        routine that `calc_seq` uses for Barlat materials (closed-form principal values of the two transformed
        deviators)."""
        if not self.barlat:
            raise AttributeError('calc_seqB: material has no Barlat parameters (plasticity(barlat=..., barlat_exp=...))')
        s, single = self._voigt(sv, 'calc_seqB')
        if s.shape[1] != 6:
            raise ValueError('calc_seqB: Voigt stress (6,) or (N,6) expected')
        seq = self._load(ana=True).seq(0, s)
        return seq[0] if single else seq

    def create_scaled_input(self, sig, epl=None, acc_strain=None, max_stress=None, flag=None, tex=None):
        """Feature vectors of the SVC yield function for stresses (material.py:2301-2346, the branch without texture and
        work-hardening features — the ones this engine evaluates): principal-stress cylinder coordinates for sdim=3,
        (deviatoric) Voigt stress / scale_seq for sdim=6.  Host-side helper for scripts; the kernels build the same
        features in registers."""
        if tex is not None or getattr(self, 'txdat', False):
            raise NotImplementedError('create_scaled_input: texture features are outside this engine')
        s, _ = self._voigt(sig, 'create_scaled_input')
        x = np.zeros((len(s), self.Ndof))
        if self.sdim == 3:
            x[:, 0] = sig_eq_j2(s) / self.scale_seq - 1.
            x[:, 1] = sig_polar_ang(s) / np.pi
        else:
            if self.dev_only:
                s = sig_dev(s)
            x[:, 0:s.shape[1]] = s / self.scale_seq
        if getattr(self, 'whdat', False):   # material.py:2342-2346
            iw = self.ind_wh
            x[:, iw:iw + self.sdim] = (0. if epl is None else np.asarray(epl, dtype=float)) / self.scale_wh
            x[:, iw + self.sdim] = 0. if acc_strain is None else acc_strain
            x[:, iw + self.sdim + 1] = (0. if max_stress is None else max_stress) / self.scale_seq
            x[:, iw + self.sdim + 2] = 0. if flag is None else flag
        return x

    def get_sflow(self, epl):
        """Scalar flow stress for a plastic strain tensor or PEEQ (material.py:974-1007)."""
        peeq = epl if type(epl) in (float, np.float64) else eps_eq(epl)
        return self.sy + peeq * self.khard

    def calc_yf(self, sig, epl=None, ana=False, pred=False, **kw):
        """Yield function: analytic ``calc_seq - sflow`` or the SVC decision function (material.py:348-412)."""
        s, single = self._voigt(sig, 'calc_yf')
        if epl is None:
            e = np.zeros((len(s), 6))
        elif type(epl) in (float, np.float64):
            e = np.tile(epl * np.array([1., -0.5, -0.5, 0., 0., 0.]), (len(s), 1))
        else:
            e = np.asarray(epl, dtype=float)
            if e.ndim == 1:
                e = np.tile(e, (len(s), 1))
        if self.ML_yf and not ana:
            f = self._load().yf(0, self._princ_rows(s), e)
            if pred:
                f = np.where(f > 0., 1., -1.)
                self.msg['yield_fct'] = 'ML_yf-predict'
            else:
                self.msg['yield_fct'] = 'ML_yf-decision-fct'
        else:
            f = self._load(ana=True).yf(0, self._princ_rows(s), e)
            self.msg['yield_fct'] = 'analytical'
        return f[0] if single else f

    def find_yloc(self, x, su, epl=None, **kw):
        """Yield function at ``sig = x[:, None] * su`` for N unit stresses at once (material.py:518-545): the objective
        that the reference's training / plotting scripts hand to ``scipy.optimize.fsolve`` to trace yield loci
        (``plot_yield_locus`` :3017, ``calc_yf`` on (N, sdim) grids); one batched GPU evaluation per call."""
        x = np.asarray(x, dtype=float)
        su = np.asarray(su, dtype=float)
        return self.calc_yf(x[:, None] * su, epl=epl)

    def find_yloc_scalar(self, x, su, epl=None, **kw):
        """scalar form of `find_yloc` (material.py:547-574): ``calc_yf(x * su)``"""
        return self.calc_yf(float(x) * np.asarray(su, dtype=float), epl=epl)

    def ML_full_yf(self, sig, epl=None, ld=None, verb=True, **kw):
        """Distance of a stress to the ML yield locus along its ray / the loading direction
        (material.py:414-516).  Accepts a single stress like the reference, or (N,6)."""
        if not self.ML_yf:
            raise AttributeError('ML_full_yf: material has no trained ML yield function')
        s = np.asarray(sig, dtype=float)
        if s.shape not in ((3,), (6,)) and not (s.ndim == 2 and s.shape[1] == 6):
            raise ValueError('Only individual stress tensors supported in material.ML_full_yf. '
                             'Shape of argument is {}'.format(s.shape))
        s, single = self._voigt(s, 'ML_full_yf')
        e = None
        if epl is not None:
            e = np.asarray(epl, dtype=float)
            if e.ndim == 1:
                e = np.tile(e, (len(s), 1))
        f, st = self._load().full_yf(0, self._princ_rows(s), e, ld)
        if verb and np.any(st != 0):
            warnings.warn('ML_full_yf: Could not bracket / locate the yield locus for %d stress(es); '
                          'conservative estimate seq-0.85*sflow returned' % int(np.sum(st != 0)))
        return f[0] if single else f

    def calc_fgrad(self, sig, epl=None, seq=None, ana=False, **kw):
        """Gradient of the yield function w.r.t. stress (material.py:704-858)."""
        s0 = np.asarray(sig, dtype=float)
        if epl is not None and np.shape(epl) != s0.shape:
            raise ValueError('Parameter sig and epl must have the same shape.')
        nd = self.sdim if self.sdim is not None else 6
        if s0.shape not in ((3,), (6,)) and not (s0.ndim == 2 and s0.shape[1] == nd):
            raise ValueError('Unknown format of stress in calc_fgrad')
        if not (self.ML_yf and not ana):
            self._no_flow_rule()
        s, single = self._voigt(s0, 'calc_fgrad')
        nout = s0.shape[-1]  # principal stresses in -> gradient w.r.t. principal stresses out
        if self.ML_yf and not ana and getattr(self, 'whdat', False):
            # the plastic strain is part of the feature vector, and the hardening modulus is read off the gradient
            # w.r.t. the plastic-strain features: mean over the points, no softening (material.py:808-814)
            e = None if epl is None else np.asarray(epl, dtype=float).reshape(len(s), -1)
            a, hk = self._load().fgrad_wh(0, s, e)
            self.khard = max(0., float(np.sum(hk)) / len(s))
            self.msg['gradient'] = 'gradient to ML_yf'
        elif self.ML_yf and not ana:
            a = self._load().fgrad(0, s)
            self.khard = 0.  # side effect of the reference (material.py:812-814, no work-hardening data)
            self.msg['gradient'] = 'gradient to ML_yf'
        else:
            # the reference's analytic form takes the deviator of the components it is GIVEN (principal stresses for (3,) /
            # (N,3) input, Voigt normals for a (6,) stress) and the equivalent stress from `seq` or calc_seq (:834-838) --
            # for a (6,) stress of a principal-stress material that is seq in sig_princ's order over the deviator of the Voigt
            # normals, not the principal-space normal that epl_dot / C_tan build (:1044-1047, `fgrad`)
            if seq is not None or (self.sdim == 3 and nout == 6):
                q = self.calc_seq(s) if seq is None else np.broadcast_to(np.asarray(seq, dtype=float).reshape(-1), (len(s),))
                a = self._load(ana=True).fgrad_seq(0, s, np.ascontiguousarray(q))
            else:
                a = self._load(ana=True).fgrad(0, s)
            h = self.hill
            if self.sdim == 6:
                self.msg['gradient'] = ('analytical, J2 isotropic, full stress' if np.all(h == 1.)
                                        else 'analytical, 6-parameter Hill, full stress')
            else:
                self.msg['gradient'] = ('analytical, J2 isotropic, princ. stress' if np.all(h == 1.)
                                        else 'analytical, 3-parameter Hill, princ. stress')
        a = a[:, :nout]
        return a[0] if single else a

    def response(self, sig, epl, deps, CV, maxit=50):
        """Elastic-predictor / plastic-corrector update of one material point (material.py:207-346).
        Returns ``fy1, sig, depl, grad_stiff``; ``msg['nsteps']`` is set as in the reference."""
        sig = np.asarray(sig, dtype=float)
        sh = sig.shape
        if sh != (6,) and sh != (3,):
            raise ValueError('Only individual stress tensors supported in material.response. '
                             'Shape of argument is {}'.format(sh))
        if sh == (3,):
            raise NotImplementedError('response: pass the Voigt stress (6,); (3,) principal input is not supported')
        self._no_flow_rule()
        maxit = int(maxit)
        if maxit < 1:
            raise ValueError('response: maxit must be >= 1')
        if self.sy is None:
            raise AttributeError('response called for a purely elastic material')
        if getattr(self, 'whdat', False):
            # Material.khard is state here: read on entry, overwritten by every gradient evaluation inside the call
            fy, so, dp, ct, ns, kout = self._load(CV).response(sig[None, :], np.asarray(epl, dtype=float)[None, :],
                                                               np.asarray(deps, dtype=float)[None, :],
                                                               khard_in=[self.khard], return_khard=True, maxit=maxit)
            self.khard = float(kout[0])
        else:
            fy, so, dp, ct, ns = self._load(CV).response(sig[None, :], np.asarray(epl, dtype=float)[None, :],
                                                         np.asarray(deps, dtype=float)[None, :], maxit=maxit)
        self.msg['nsteps'] = int(ns[0])
        return fy[0], so[0], dp[0], ct[0].reshape(6, 6)

    def response_batch(self, sig, epl, deps, CV):
        """``response`` on (N,6) arrays in one launch (extension; same numbers point by point)."""
        fy, so, dp, ct, ns = self._load(CV).response(sig, epl, deps)
        return fy, so, dp, ct.reshape(-1, 6, 6), ns

    def epl_dot(self, sig, epl, Cel, deps, **kw):
        """Plastic strain increment relaxing the stress to the yield locus (material.py:1009-1055)."""
        sig = np.asarray(sig, dtype=float)
        Cel = np.asarray(Cel, dtype=float)
        deps = np.asarray(deps, dtype=float)
        yfun = self.calc_yf(sig + Cel @ deps, epl=epl)
        if yfun <= yf_tolerance:
            return np.zeros(6)
        a = self._flow_normal(sig, epl)
        hh = a @ Cel @ a + self.khard
        return (a @ Cel @ deps / hh) * a

    def _flow_normal(self, sig, epl=None):
        """Normal of the flow rule as epl_dot / C_tan take it: calc_fgrad of the Voigt stress for sdim = 6; for sdim = 3 the
        gradient w.r.t. the PRINCIPAL stresses in the normal Voigt rows, shear rows zero (material.py:1044-1047, 1079-1081) --
        not calc_fgrad's form for a (6,) stress of such a material (see there)."""
        if self.sdim == 3 and not (self.tresca or self.barlat):
            s, _ = self._voigt(sig, 'calc_fgrad')
            if not self.ML_yf:
                self._no_flow_rule()
            return self._load(ana=not self.ML_yf).fgrad(0, self._princ_rows(s))[0]
        return self.calc_fgrad(sig) if epl is None else self.calc_fgrad(sig, epl=np.asarray(epl, dtype=float))

    def C_tan(self, sig, Cel, epl=None):
        """Continuum tangent stiffness (material.py:1057-1086)."""
        Cel = np.asarray(Cel, dtype=float)
        # the reference evaluates calc_fgrad(sig, epl=epl) with zeros when epl is None (material.py:1076-1082): for a
        # work-hardening SVC the plastic strain is part of the feature vector and the call sets self.khard used below
        epl = np.zeros(self.sdim) if epl is None else np.asarray(epl, dtype=float)
        a = self._flow_normal(np.asarray(sig, dtype=float), epl)
        ca = Cel @ a
        return Cel - np.outer(ca, ca) / (a @ ca + self.khard)

    # ------------------------------------------------------------------ harness
    def calc_properties(self, size=2, Nel=2, verb=False, eps=0.005, min_step=None, sigeps=False,
                        load_cases=['stx', 'sty', 'et2', 'ect']):
        """Stress-strain curves of a 2x2 plane-stress model under four load cases
        (material.py:3062-3166); the harness of the reference's plasticity tests."""
        from .model import Model

        def calc_strength(vbc1, nbc1, vbc2, nbc2, sel):
            fe = Model(dim=2, planestress=True)
            fe.geom([size], LY=size)
            fe.assign([self])
            fe.bcleft(0.)
            fe.bcbot(0.)
            fe.bcright(vbc1, nbc1)
            fe.bctop(vbc2, nbc2)
            fe.mesh(NX=Nel, NY=Nel)
            fe.solve(verb=verb, min_step=min_step)
            seq = self.calc_seq(fe.sgl)
            eeq = eps_eq(fe.egl)
            peeq = eps_eq(fe.epgl)
            iys = np.nonzero(peeq < 1.e-2)
            self.prop[sel].update(ys=seq[iys[0][-1]], seq=seq, eeq=eeq, peeq=peeq)
            seq = sig_eq_j2(fe.sgl)
            iys = np.nonzero(peeq < 1.e-6)
            self.propJ2[sel].update(ys=seq[iys[0][-1]], seq=seq, eeq=eeq, peeq=peeq)
            if sigeps:
                self.sigeps[sel].update(sig=fe.sgl, eps=fe.egl, epl=fe.epgl)

        cases = {'stx': (eps * size, 'disp', 0., 'force', '-r', 'uniax-x'),
                 'sty': (0., 'force', eps * size, 'disp', '-b', 'uniax-y'),
                 'et2': (0.4 * eps * size, 'disp', 0.4 * eps * size, 'disp', '-k', 'equibiax'),
                 'ect': (-0.8 * eps * size, 'disp', 0.8 * eps * size, 'disp', '-m', 'shear')}
        for case in load_cases:
            if case not in cases:
                warnings.warn('calc_properties: Load case not supported: {}'.format(case))
                continue
            v1, n1, v2, n2, style, label = cases[case]
            calc_strength(v1, n1, v2, n2, case)
            self.prop[case]['style'] = style
            self.prop[case]['name'] = label
