#!/usr/bin/env python3
"""M independent ogre.obj-size solves on one GPU (bench.py's multi_mesh leg on its own): python tools/multi_mesh.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
print(json.dumps(B.multi_mesh_leg(smg, mesh, torch, dev), indent=1))
