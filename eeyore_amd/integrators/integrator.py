class Integrator:
    """Base class for integration (eeyore/integrators/integrator.py)."""

    def integrate(self):
        raise NotImplementedError
