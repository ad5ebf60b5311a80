"""CPU: examples/smg_eigen_adapter.cpp -- the reference's Eigen-typed functions on libsmg -- must at least be
well-formed C++.  Eigen is not in the image, so it is compiled against tests/mock_eigen: a ~100-line stand-in with the members the
adapter touches.  This is a syntax / interface-subset check, clearly NOT the reference or Eigen compiled; the GPU suite
(tests/test_gpu_cpp_api.py) then runs the adapter on the mock containers."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_adapter_is_well_formed_against_the_mock():
    for src in ("smg_eigen_adapter.cpp", "adapter_check.cpp"):
        r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-DSMG_ADAPTER_MOCK",
                            "-I" + os.path.join(ROOT, "tests", "mock_eigen"), "-I" + os.path.join(ROOT, "include"),
                            os.path.join(ROOT, "examples", src)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]


def test_adapter_defines_every_overload_of_the_reference_headers():
    """8 solve overloads + 2 precompute overloads + the six mg_VCycle.h pieces (src/min_quad_with_fixed_mg.h:32-113,
    src/mg_VCycle.h:22-76), instantiated for column vectors and dense blocks like src/min_quad_with_fixed_mg.cpp:363-373."""
    txt = open(os.path.join(ROOT, "examples", "smg_eigen_adapter.cpp")).read()
    assert txt.count("bool min_quad_with_fixed_mg_solve(") == 6
    assert txt.count("void min_quad_with_fixed_mg_precompute(") == 2
    for fn in ("void mg_VCycle(", "void A(", "void restrict(", "void prolong(", "void relax(", "void coarseSolve("):
        assert txt.count(fn) == 1, fn
    assert "SMG_INST_SOLVE(Eigen::VectorXd)" in txt and "SMG_INST_SOLVE(Eigen::MatrixXd)" in txt
    mock = sum(len(open(os.path.join(dp, f)).read().splitlines()) for dp, _, fs in os.walk(os.path.join(ROOT, "tests", "mock_eigen")) for f in fs)
    assert mock <= 130, "the mock is meant to stay a handful of members, not grow into a library"
