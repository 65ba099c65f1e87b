"""eeyore_amd: MI355X-native multi-chain MCMC engine for Bayesian MLPs behind eeyore's sampler/model surface.

Sub-packages mirror the reference layout (papamarkou/eeyore): constants, datasets, models, samplers, chains,
kernels, tuners, integrators, stats.  The hot path (MLP log-target, its gradient, the HMC/MALA/MH step) runs in
the HIP library ``eeyore_amd/lib/libeeyore_amd.so`` through the C ABI of ``include/eeyore_amd.h``.
"""
__version__ = "0.1.0"
