"""CPU: the C-ABI library loads, exports every symbol include/smg.h declares, and its compute entry points fail
loudly (SMG_ERR_NO_DEVICE) instead of falling back to a CPU path when there is no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from problems import subdiv_problem

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "smg.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(smg_[a-zA-Z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported(smg_mod):
    from surface_multigrid_code_amd import _lib
    L = _lib.load()
    names = _declared()
    assert len(names) > 40
    for n in names:
        assert hasattr(L, n), "libsmg.so does not export %s declared in include/smg.h" % n
    # and the python binding covers all of them
    assert sorted(_lib.exported_symbols()) == names


def test_version_and_defaults(smg_mod):
    from surface_multigrid_code_amd import _lib
    L = _lib.load()
    import re
    declared = int(re.search(r"#define\s+SMG_VERSION\s+(\d+)", open(os.path.join(ROOT, "include", "smg.h")).read()).group(1))
    assert L.smg_version() == declared >= 210
    o = _lib.SolveOptsC()
    L.smg_solve_opts_default(C.byref(o))
    # reference defaults: tol 1e-3, maxIter 20, pre = post = 2 (src/min_quad_with_fixed_mg.cpp:63,77,102-103)
    assert (o.tol, o.max_iter, o.pre, o.post) == (1e-3, 20, 2, 2)


def test_no_oracle_in_product():
    """The product path must not import/link the oracle."""
    pkg = os.path.join(ROOT, "surface_multigrid_code_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "smg_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


@pytest.mark.skipif(os.environ.get("SMG_EXPECT_GPU") == "1", reason="GPU box")
def test_compute_fails_loudly_without_gpu(smg_mod):
    smg = smg_mod
    if smg._lib.load().smg_device_count() > 0:
        pytest.skip("a GPU is present")
    p = subdiv_problem(kind="mcf", k=1, n_sub=1)
    mg = smg.Hierarchy.from_prolongs(p["Ps"])
    with pytest.raises(smg.SmgError) as e:
        mg.precompute(p["A"])
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)
    with pytest.raises(smg.SmgError):
        mg.solve(p["RHS"], p["z0"])
    with pytest.raises(smg.SmgError):
        mg.A(0, p["z0"])
