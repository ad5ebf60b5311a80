#!/usr/bin/env python3
"""Convert the reference's OBJ *data files* (meshes/) into compact binary fixtures.

Run in the build container only (needs /root/reference).  Output: tests/golden/meshes/*.smgm

Format (little endian):  b"SMGM" | u32 version=1 | i32 nV | i32 nF | f64 V[nV*3] | i32 F[nF*3]
Only `v` and `f` records are kept (what igl::read_triangle_mesh returns to the callers,
03_mg_solver/main.cpp:29); normals / texcoords are dropped; faces are 0-based.
"""
import os, struct, sys
import numpy as np

SRC = "/root/reference/meshes"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "meshes")
NAMES = ["bunny", "bunny_15K_init", "ogre", "ogre_sim", "hilbert_cube_known"]


def read_obj(path):
    V, F = [], []
    with open(path) as f:
        for line in f:
            if line.startswith("v "):
                p = line.split()
                V.append((float(p[1]), float(p[2]), float(p[3])))
            elif line.startswith("f "):
                p = line.split()[1:]
                idx = [int(t.split("/")[0]) - 1 for t in p]
                for k in range(1, len(idx) - 1):      # fan-triangulate (all inputs are triangles)
                    F.append((idx[0], idx[k], idx[k + 1]))
    return np.asarray(V, dtype=np.float64), np.asarray(F, dtype=np.int32)


def write_smgm(path, V, F):
    with open(path, "wb") as f:
        f.write(b"SMGM")
        f.write(struct.pack("<Iii", 1, V.shape[0], F.shape[0]))
        f.write(np.ascontiguousarray(V, dtype="<f8").tobytes())
        f.write(np.ascontiguousarray(F, dtype="<i4").tobytes())


if __name__ == "__main__":
    os.makedirs(DST, exist_ok=True)
    for n in NAMES:
        V, F = read_obj(os.path.join(SRC, n + ".obj"))
        write_smgm(os.path.join(DST, n + ".smgm"), V, F)
        print(n, V.shape, F.shape)
