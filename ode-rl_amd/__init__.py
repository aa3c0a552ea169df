"""MI355X-native Neural-ODE hot path of jithendaraa/ODE-RL.

Public surface = the reference's: `odeint`, `DiffEqSolver`, `ODEFunc`, `create_convnet`.
"""
from . import _lib  # noqa: F401
from .odeint import odeint, last_stats  # noqa: F401
from .autograd import odeint_adjoint, last_adjoint_stats  # noqa: F401
from .hip_ops import set_compute_dtype, current_compute_dtype, set_async_dopri5, collect_pending_solves  # noqa: F401
from .helpers.utils import create_convnet  # noqa: F401
from .modules.DiffEqSolver import DiffEqSolver, ODEFunc  # noqa: F401
from .modules.ConvGRUCell import ConvGRUCell  # noqa: F401
from .modules.ODEConvGRUCell import ODEConvGRUCell  # noqa: F401

__all__ = ["odeint", "odeint_adjoint", "DiffEqSolver", "ODEFunc", "create_convnet", "ConvGRUCell", "ODEConvGRUCell"]
