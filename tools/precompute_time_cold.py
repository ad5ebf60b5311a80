#!/usr/bin/env python3
"""First smg_precompute of a process (HIP not yet initialised by anyone), C3, then a second handle in the same process.  SMG_TIMING=1 for the stages."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
mg, A, Mb, Vf, Ff, label, t_setup = B.build_workload("C3", smg, mesh)
t0 = time.time(); mg.precompute(A); print("first precompute of the process: %.3f s" % (time.time() - t0), flush=True)
mg2, A2, *_ = B.build_workload("C3", smg, mesh)
t0 = time.time(); mg2.precompute(A2); print("first precompute of a second handle: %.3f s" % (time.time() - t0), flush=True)
