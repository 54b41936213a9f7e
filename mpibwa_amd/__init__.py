"""mpibwa_amd — MI355X-native BWA-MEM hot path behind mpiBWA's mem_process_seqs() boundary.

The product is the C-ABI shared library libmpibwa_amd.so (include/mpibwa_amd.h);
this package only builds it, loads it with ctypes and mirrors the reference's
call interface for tests and bench.py.
"""
from . import abi  # noqa: F401
from .api import Engine, load_library, build_index  # noqa: F401
