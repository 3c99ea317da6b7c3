"""gym_electric_motor_amd -- MI355X-native batched physical-system stepper for gym-electric-motor (GEM).

One hot path only: `SCMLSystem.simulate()` / `reset()` for N independent env instances, in hand-written HIP
kernels behind GEM's `PhysicalSystem` plugin surface (include/gemx.h).
"""
from . import _lib  # noqa: F401
from .components import (  # noqa: F401
    ConstantSpeedLoad,
    ContB6BridgeConverter,
    ContFourQuadrantConverter,
    ContMultiConverter,
    DcExternallyExcitedMotor,
    DcPermanentlyExcitedMotor,
    DcSeriesMotor,
    DcShuntMotor,
    DoublyFedInductionMotor,
    DormandPrince5Solver,
    ScipyOdeSolver,
    EulerSolver,
    ExternallyExcitedSynchronousMotor,
    FiniteB6BridgeConverter,
    FiniteFourQuadrantConverter,
    FiniteMultiConverter,
    IdealVoltageSupply,
    PermanentMagnetSynchronousMotor,
    PolynomialStaticLoad,
    RCVoltageSupply,
    RK4Solver,
    SquirrelCageInductionMotor,
    SynchronousReluctanceMotor,
)
from .envs import BatchedElectricMotorEnv, default_ode_solver, make  # noqa: F401
from .reference_generators import BatchedWienerProcessReferenceGenerator  # noqa: F401
from .physical_system_wrappers import DeadTimeProcessor, DqToAbcActionProcessor  # noqa: F401
from .physical_systems import (  # noqa: F401
    BatchedDcMotorSystem,
    BatchedDoublyFedInductionMotorSystem,
    BatchedExternallyExcitedSynchronousMotorSystem,
    BatchedSCMLSystem,
    BatchedSquirrelCageInductionMotorSystem,
    BatchedSynchronousMotorSystem,
    LimitConstraint,
    SquaredConstraint,
)

__version__ = "0.1.0"
