"""surface_multigrid_code_amd -- MI355X-native surface multigrid V-cycle (libsmg) and its Python host mirror.

The product is the C-ABI shared library `lib/libsmg.so` (include/smg.h): C++ host code + hand-written gfx950 HIP
kernels.  This package is a thin ctypes mirror of the reference's operator API for that path
(`mg_precompute` -> `min_quad_with_fixed_mg_precompute` -> `min_quad_with_fixed_mg_solve` / `mg_VCycle`),
used by tests and bench.py.  There is NO CPU fallback: compute calls raise if the library or a GPU is missing.
"""
from .api import (Hierarchy, SmgError, mg_precompute, mg_precompute_block, mg_precompute_subdiv, min_quad_with_fixed_mg_precompute,
                  min_quad_with_fixed_mg_solve, mg_VCycle, SolveOpts, query_coarse_to_fine, query_fine_to_coarse)
from . import mesh

__all__ = ["Hierarchy", "SmgError", "mg_precompute", "mg_precompute_block", "mg_precompute_subdiv", "min_quad_with_fixed_mg_precompute",
           "min_quad_with_fixed_mg_solve", "mg_VCycle", "SolveOpts", "query_coarse_to_fine", "query_fine_to_coarse", "mesh"]
