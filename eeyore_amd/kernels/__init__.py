from .proposal import Kernel, NormalizedKernel, NormalKernel
