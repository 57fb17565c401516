"""varpro_amd -- MI355X-native batched variable projection (the hot path of geo-ant/varpro).

Host-side mirror of the reference's public surface over hand-written HIP kernels for gfx950:

    SeparableModelBuilder / SeparableModel         (src/model/)
    SeparableProblemBuilder / SeparableProblem     (src/problem.rs, src/problem/builder.rs)
    LevMarSolver / FitResult / LevenbergMarquardt  (src/solvers/levmar/mod.rs, src/fit.rs)
    BatchProblem                                   (new: B independent problems per launch)

There is no CPU fallback: every compute call runs on the GPU through ``lib/libvarpro_hip.so``
(C ABI in include/varpro_hip.h) or raises.
"""
from ._lib import VarproHipError, VarproHipUnavailable, device_count, load as load_library  # noqa: F401
from .batch import BatchProblem, LevenbergMarquardt, REPORT_DTYPE  # noqa: F401
from .pipeline import FitPipeline  # noqa: F401
from .model import (ClosureModel, ExternalModel, ModelBuildError, ModelError, SeparableModel,  # noqa: F401
                    SeparableModelBuilder, basis, multi_exponential_model)
from .problem import SeparableProblem, SeparableProblemBuilder, SeparableProblemBuilderError  # noqa: F401
from .solver import (FitError, FitResult, FitStatistics, LevMarSolver, MinimizationReport,  # noqa: F401
                     TerminationReason)  # noqa: F401

__version__ = "0.1.0"
