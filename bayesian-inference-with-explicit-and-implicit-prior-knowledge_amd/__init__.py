"""MI355X-native Particle-Gibbs-with-Ancestor-Sampling engine.

Host-side mirror of the reference's call surface for its particle-filter hot path
(reference src/PGAS.py, src/Filtering.py, src/BasisFunctions.py, src/BayesianInferrence.py) over
hand-written HIP kernels in ``libpgas_hip.so`` (C ABI: include/pgas_hip.h).

Import as ``pgas_amd`` (the shim pgas_amd.py at the repository root maps that name onto this
directory, whose own name is not a Python identifier).
"""
from . import random  # noqa: F401
from .BasisFunctions import generate_Hilbert_BasisFunction  # noqa: F401
from .BayesianInferrence import (  # noqa: F401
    prior_mniw_2naturalPara,
    prior_mniw_2naturalPara_inv,
    prior_mniw_calcStatistics,
    prior_mniw_drawPred,
    prior_mniw_log_base_measure,
    prior_mniw_mean,
    prior_mniw_Predictive,
)
from .descriptors import BasisMap, GaussianLikelihood, HilbertBasis  # noqa: F401
from .Filtering import reconstruct_trajectory, systematic_SISR  # noqa: F401
from .Algorithm1 import Algorithm1  # noqa: F401
from .Algorithm3 import Algorithm3  # noqa: F401
from .Algorithm2 import Algorithm2  # noqa: F401
from .PGAS import PGAS, condSequentialMonteCarlo  # noqa: F401
from .StateSpaceModel import StateSpaceModel, SymbolicStateSpaceModel  # noqa: F401

__all__ = [
    "PGAS",
    "Algorithm1",
    "Algorithm2",
    "Algorithm3",
    "StateSpaceModel", "SymbolicStateSpaceModel",
    "condSequentialMonteCarlo",
    "generate_Hilbert_BasisFunction",
    "HilbertBasis",
    "BasisMap",
    "GaussianLikelihood",
    "prior_mniw_2naturalPara",
    "prior_mniw_2naturalPara_inv",
    "prior_mniw_calcStatistics",
    "prior_mniw_mean",
    "prior_mniw_Predictive",
    "prior_mniw_drawPred",
    "prior_mniw_log_base_measure",
    "random",
    "systematic_SISR",
    "reconstruct_trajectory",
]
