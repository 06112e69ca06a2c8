"""cymf_amd -- MI355X-native drop-in for the cymf.BPR / WMF / RelMF / GloVe / ExpoMF class surface
(minatosato/cymf, cymf/__init__.py:1-7).  Python host -> ctypes -> libcymf_hip.so (HIP, gfx950).
There is no CPU fallback: the classes raise if the HIP library or a gfx950 device is missing."""
from .bpr import BPR
from .wmf import WMF
from .relmf import RelMF
from .glove import GloVe
from .expomf import ExpoMF
from . import synthetic
from . import dataset
from .evaluator import Evaluator, AverageOverAllEvaluator, AoaEvaluator, UnbiasedEvaluator

__version__ = "0.1.0"
__all__ = ["BPR", "WMF", "RelMF", "GloVe", "ExpoMF", "Evaluator", "AverageOverAllEvaluator", "AoaEvaluator",
           "UnbiasedEvaluator", "synthetic", "dataset"]
