"""CPU: the AddressSanitizer + UndefinedBehaviorSanitizer lane for the host C++ ("Race detection / sanitizers").
Builds lib/libsmg_asan.so (surface_multigrid_code_amd/build.py: the host translation units through g++ -fsanitize=address,undefined,
the device object unchanged) and runs, in a child python with libasan preloaded and SMG_LIB pointing at it,
  * the host-logic and ABI tests (precompute slices / Galerkin products / orderings / decimator invariants / malformed inputs / save+load),
  * tools/fuzz_host.py: random meshes through all three decimators and the host half of the precompute.
Any sanitizer report fails the test."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout):
    from surface_multigrid_code_amd import build as smg_build
    smg_build.build_sanitized()
    env = smg_build.sanitizer_env()
    env["SMG_HOST_THREADS"] = "4"
    r = subprocess.run([sys.executable] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    out = r.stdout + r.stderr
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert r.returncode == 0, out[-4000:]
    return out


def test_sanitized_library_is_the_one_loaded():
    out = _run(["-c", "import surface_multigrid_code_amd as s; from surface_multigrid_code_amd import _lib; print('LIB', _lib.LIB_PATH, _lib.load().smg_version())"], 300)
    import re
    declared = re.search(r"#define\s+SMG_VERSION\s+(\d+)", open(os.path.join(ROOT, "include", "smg.h")).read()).group(1)
    assert "libsmg_asan.so " + declared in out


def test_host_logic_and_abi_under_asan_ubsan():
    out = _run(["-m", "pytest", "tests/test_host_logic.py", "tests/test_abi.py", "-x", "-q", "-p", "no:cacheprovider"], 1500)
    assert " passed" in out and " failed" not in out


def test_host_fuzz_under_asan_ubsan():
    out = _run(["tools/fuzz_host.py", "10", "0"], 1500)
    assert "FUZZ_HOST OK" in out


def test_threaded_host_half_of_the_precompute_under_tsan():
    """ThreadSanitizer over the host half of a first smg_precompute (its own thread beside the caller's, the early locality-order and colouring
    threads, the persistent pool, the hand-over object): tests/tsan_precompute_driver.cpp, three rounds on a 77 k-row, 4-level system, two of them also under the
    decimated hierarchy smg_mg_precompute builds for the same mesh (level 0 coloured from scratch on its own thread).
    Without a device the call ends with SMG_ERR_NO_DEVICE after the host half has run -- which is what this lane is about."""
    from surface_multigrid_code_amd import build as smg_build
    exe = smg_build.build_tsan()
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0", SMG_HOST_THREADS="6",
               HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")     # (on a box with a GPU: keep the runtime's own threads out of the report)
    r = subprocess.run([exe], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    out = r.stdout + r.stderr
    if "unexpected memory mapping" in out:       # kernels with more address-space randomisation than this libtsan knows: run without it
        import shutil
        import pytest
        if not shutil.which("setarch"):
            pytest.skip("ThreadSanitizer cannot map its shadow memory on this kernel and setarch is not available")
        r = subprocess.run(["setarch", os.uname().machine, "-R", exe], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        out = r.stdout + r.stderr
        if "unexpected memory mapping" in out:
            pytest.skip("ThreadSanitizer cannot map its shadow memory on this kernel")
    assert "ThreadSanitizer" not in out, out[-6000:]
    assert out.count("levels, precompute rc = ") == 3 and out.count("decimated precompute rc = ") == 2 and r.returncode == 0, out[-3000:]
