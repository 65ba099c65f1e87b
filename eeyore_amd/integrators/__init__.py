from .integrator import Integrator
from .mcintegrator import MCIntegrator
