import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
torch.cuda.init(); torch.zeros(1, device="cuda")
import bench as B
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
mg, A, Mb, Vf, Ff, label, t_setup = B.build_workload("C3", smg, mesh)
t0 = time.time(); mg.precompute(A); print("precompute (HIP already up): %.3f s" % (time.time() - t0))
