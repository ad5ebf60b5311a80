#!/usr/bin/env python3
"""Reference-derived golden vectors for the hierarchy builder (row f-1 of SURVEY.md section 8).

The reference checks in the OUTPUT of its 08_subdiv_remesh example (08_subdiv_remesh/output_s0.obj, _s1.obj, _s2.obj): bunny.obj
decimated by mid-point collapse to 500 faces (SSP_decimate, dec_type 1), the coarse mesh mid-point-upsampled 0 / 1 / 2 times, and every
vertex of the upsampled mesh carried back onto the input surface through the bijection of the successive self-parameterisation
(query_coarse_to_fine; 08_subdiv_remesh/main.cpp:131-166).  These positions -- written by the reference itself with 15 significant
digits -- are data, copied here as a fixture (vertices and faces of the three files):

    tests/golden/bunny_remesh_500.npz   s0_V (261 x 3), s0_F (499 x 3), s1_V (1020 x 3), s1_F (1996 x 3), s2_V (4035 x 3), s2_F (7984 x 3)

tests/test_host_logic.py::test_kat_subdiv_remesh_outputs_of_the_reference maps the same points through libsmg's decimator and
smg_query_coarse_to_fine and compares.  Run in the build container (needs /root/reference):  python tests/golden/make_remesh_golden.py
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/08_subdiv_remesh"


def read_obj(path):
    V, F = [], []
    for ln in open(path):
        t = ln.split()
        if not t:
            continue
        if t[0] == "v":
            V.append([float(x) for x in t[1:4]])
        elif t[0] == "f":
            F.append([int(x.split("/")[0]) - 1 for x in t[1:4]])
    return np.array(V, dtype=np.float64), np.array(F, dtype=np.int32)


if __name__ == "__main__":
    out = {}
    for k in (0, 1, 2):
        V, F = read_obj(os.path.join(REF, "output_s%d.obj" % k))
        out["s%d_V" % k] = V
        out["s%d_F" % k] = F
        print("output_s%d.obj: %d vertices, %d faces" % (k, V.shape[0], F.shape[0]))
    np.savez_compressed(os.path.join(HERE, "bunny_remesh_500.npz"), **out)
