"""MI355X-native batched linear interpolation (gfx950 HIP behind a C ABI).

Public host API: see api.py; the C ABI is include/mi355_interp.h and the
Armadillo-facing C++ operator signatures are include/mi355_arma.hpp.
Importing this package does not touch the GPU; the first Context() does.
"""
from ._lib import MiError, lib_path, load  # noqa: F401
from .api import (  # noqa: F401
    MATH_EXACT, MATH_FAST, Context, EventDrivenMap, Grid1, Grid2, Group, Timer, default_edm_params, interp1,
    masked_mean, restrict, restrict_mean, shard_bounds,
)
