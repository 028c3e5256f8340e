"""sopht_mpi_amd: the sopht-mpi Eulerian flow-step hot path on MI355X.

Same public names as ``sopht_mpi`` (simulator / numeric / utils), backed by
libsophtmi.so (hand-written HIP kernels, rocFFT, RCCL).  No CPU fallback.
"""
from . import _lib  # noqa: F401  (fails loudly when the HIP library is missing)
from . import utils, numeric, simulator  # noqa: F401

__version__ = "0.1.0"
