"""
Host-side mirror of the reference's functional-map library for the matching hot path
(densematcher/pyFM in the reference): same names, argument order and error behaviour, NumPy in /
NumPy out; the arithmetic runs in libdensematch (HIP, gfx950) through densematcher_amd.engine.
"""
from . import spectral, refine, signatures  # noqa: F401
from .functional import FunctionalMapping  # noqa: F401
from .mesh import TriMesh  # noqa: F401
