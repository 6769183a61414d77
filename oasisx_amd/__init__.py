"""oasisx_amd -- the per-time-step IPCS hot path of oasisx on AMD Instinct MI355X (gfx950).

Same names as reference src/oasisx/__init__.py:12-18; the compute path is the HIP library
``liboasisx_hip.so`` (see include/oasisx_hip.h) and fails loudly when it is missing.
"""
import logging

from . import fem, io, mesh  # noqa: F401
from .bcs import DirichletBC, LocatorMethod, PressureBC
from .fracstep import FractionalStep_AB_CN
from .function import Projector
from .ksp import KSPSolver  # noqa: F401

logging.basicConfig()
logger = logging.getLogger("oasisx")
logging.captureWarnings(capture=True)

__all__ = [
    "Projector",
    "FractionalStep_AB_CN",
    "DirichletBC",
    "LocatorMethod",
    "PressureBC",
]
