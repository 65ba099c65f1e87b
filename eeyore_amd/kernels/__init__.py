from .kernel import Kernel
from .normalized_kernel import NormalizedKernel
from .normal_kernel import NormalKernel
