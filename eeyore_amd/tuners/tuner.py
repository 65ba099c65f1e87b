class Tuner:
    """Base class for tuners (eeyore/tuners/tuner.py)."""

    def tune(sampler):
        raise NotImplementedError
