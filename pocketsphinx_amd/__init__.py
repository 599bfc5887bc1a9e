"""pocketsphinx_amd -- MI355X-native acoustic scoring + Viterbi step for PocketSphinx.

The product is the C-ABI library ``libpsgpu.so`` (HIP kernels for gfx950,
declared in ``include/psgpu.h``).  This package is the thin Python host side
used by the tests and the benchmark: a ctypes binding (``capi``) and
operator-level mirrors of the reference interfaces (``ptm.PtmMgau``).

There is no CPU fallback: importing works anywhere, but every compute call
raises ``PsgpuError`` unless the HIP library is built and a gfx950 device is
present.
"""
from .capi import PsgpuError, lib, build_library, LIB_PATH  # noqa: F401
from .ptm import PtmModel, PtmMgau, PtmState  # noqa: F401
from .hmm import HmmContext, HMM_REC  # noqa: F401
from .semi import SemiMgau  # noqa: F401
from .ms import MsMgau  # noqa: F401
from .feat import dynfeat_1s_c_d_dd, FeatType  # noqa: F401
from .fe import FrontEnd  # noqa: F401
from .search import FwdtreeSearch, backtrace  # noqa: F401
from .lm import NGramTrieLM  # noqa: F401
from .flat import FwdflatSearch  # noqa: F401
from .decode import DecodePipeline  # noqa: F401

__all__ = ["PsgpuError", "lib", "build_library", "LIB_PATH", "PtmModel", "PtmMgau", "PtmState", "HmmContext", "HMM_REC", "SemiMgau", "MsMgau", "dynfeat_1s_c_d_dd", "FrontEnd", "FwdtreeSearch", "backtrace", "NGramTrieLM", "FwdflatSearch", "DecodePipeline"]
