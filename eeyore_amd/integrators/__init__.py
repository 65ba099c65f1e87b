from .mcintegrator import Integrator, MCIntegrator
