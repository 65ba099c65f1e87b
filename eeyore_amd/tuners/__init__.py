from .dual_averaging import HMCDATuner, Tuner
