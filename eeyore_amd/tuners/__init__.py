from .tuner import Tuner
from .hmcda_tuner import HMCDATuner
