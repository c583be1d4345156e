"""pylabfea_amd — MI355X-native engine for pyLabFEA's elastic-plastic hot path.

Drop-in for the path ``Model.solve() -> Element -> Material.response()`` of pyLabFEA v4.4.2:
the per-element return mapping, the B^T D B assembly and the linear solve run as hand-written
HIP kernels (gfx950) in ``libplfx.so`` behind the reference's own ``Model`` / ``Material`` API.
See DESIGN.md for the scope table and INTEGRATION.md for the C-ABI.
"""
from .basic import Strain, Stress, eps_eq, sig_dev, sig_eq_j2, sig_polar_ang, sig_princ, yf_tolerance
from .material import Material
from .model import Model
from ._dist import host_transport

__version__ = '0.1.0'
__all__ = ['Material', 'Model', 'host_transport', 'Stress', 'Strain', 'eps_eq', 'sig_dev', 'sig_eq_j2', 'sig_polar_ang', 'sig_princ',
           'yf_tolerance']
