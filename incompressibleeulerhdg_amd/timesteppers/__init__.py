from .common import IncompressibleEuler  # noqa: F401
from .hdg_imex import *  # noqa: F401,F403
from .hdg_implicit import IncompressibleEulerHDGImplicit  # noqa: F401
