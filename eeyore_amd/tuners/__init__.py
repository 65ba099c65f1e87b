from .dual_averaging import HMCDATuner, PerChainDATuner, Tuner
