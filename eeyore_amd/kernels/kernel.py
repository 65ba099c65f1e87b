class Kernel:
    """Base class for kernels (eeyore/kernels/kernel.py:4-8)."""

    def k(self, x1, x2):
        raise NotImplementedError
