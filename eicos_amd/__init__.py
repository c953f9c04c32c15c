"""eicos_amd -- MI355X-native batched SOCP interior-point solver (EiCOS-compatible hot path).

The product is the C-ABI shared library eicos_amd/libeicos_amd.so (sources in
eicos_amd/csrc, ABI in include/eicos_amd.h).  This Python package is only a thin ctypes
mirror of that ABI for tests and bench.py.  It never imports anything from oracle/.
"""
from .binding import BatchSolver, MultiBatchSolver, PinnedArray, host_register, host_unregister, Info, build_library, library_path, device_count, set_arithmetic_profile  # noqa: F401
from .problem_io import Pattern, Values, read_ecos_header, read_epb, read_problem, write_ecos_header, write_epb  # noqa: F401
