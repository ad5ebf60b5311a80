"""Build libsmg.so (the C-ABI shared library: host C++ + hand-written gfx950 HIP kernels) in-tree.

    python -m surface_multigrid_code_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  -ffp-contract=off keeps multiply and add separate on host and
device so per-row sums are bit-identical to the reference's (FMA-free) Eigen CPU kernels.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libsmg.so")
SOURCES = ["smg_device.hip", "smg_bsr3_device.hip", "smg_tiled_device.hip", "smg_coarse_device.hip", "smg_bgs_device.hip", "smg_wgs_device.hip", "smg_union_device.hip", "smg_schur_device.hip", "smg_bgs.cpp", "smg_wgs.cpp", "smg_union.cpp", "smg_schur.cpp", "smg_bsr3.cpp", "smg_tiled.cpp", "smg_coarse.cpp", "smg_capi.cpp", "smg_precompute.cpp", "smg_cycle.cpp", "smg_hierarchy_io.cpp", "smg_sparse.cpp", "smg_mesh.cpp",
           "smg_order.cpp", "smg_decimate.cpp"]
HEADERS = ["smg_device.hpp", "smg_hier.hpp", "smg_internal.hpp", "smg_device_inl.hpp", "smg_gj_inl.hpp", "smg_schur.hpp", "smg_bsr3.hpp", "smg_tiled.hpp", "smg_coarse.hpp", "smg_bgs.hpp", "smg_wgs.hpp", "smg_sparse.hpp", "smg_mesh.hpp", "smg_order.hpp",
           os.path.join("..", "..", "include", "smg.h")]
# -amdgpu-kernarg-preload-count: the leading scalar / pointer kernel arguments arrive in SGPRs with the wave (k_sell orders its
# arguments for this: its first panel loads need no kernarg read at all).  SMG_KERNARG_PRELOAD=0 builds without it (A/B).
PRELOAD = [] if os.environ.get("SMG_KERNARG_PRELOAD", "1") == "0" else ["-mllvm", "-amdgpu-kernarg-preload-count=16"]
FLAGS = (["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-result"] + PRELOAD +
         os.environ.get("SMG_EXTRA_FLAGS", "").split())


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(
                [os.path.getmtime(src)] + [os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS]):
            continue
        cmd = [_hipcc()] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("build failed: " + " ".join(cmd))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


# ---- sanitizer lane ("-fsanitize=address,undefined CPU test config") ------------------------------------------
# The host translation units -- decimator, colouring / ordering, sparse algebra, mesh numerics, the C ABI: the pointer-heavy
# code -- compiled by g++ with AddressSanitizer + UndefinedBehaviorSanitizer; the device file keeps its normal hipcc object (device
# code cannot carry host sanitizer instrumentation).  Result: lib/libsmg_asan.so, loaded through SMG_LIB by tests/test_sanitized_host.py
# with LD_PRELOAD=libasan (python itself is not instrumented).
ASAN_LIB = os.path.join(LIBDIR, "libsmg_asan.so")
ASAN_FLAGS = ["-std=c++17", "-O1", "-g", "-fPIC", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined",
              "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-w"]


def build_sanitized(verbose=False):
    build(verbose=verbose)                                   # the device object comes from the regular build
    objdir = os.path.join(LIBDIR, "obj_asan")
    os.makedirs(objdir, exist_ok=True)
    host = [s for s in SOURCES if s.endswith(".cpp")]
    objs, procs = [os.path.join(LIBDIR, "obj", os.path.splitext(s)[0] + ".o") for s in SOURCES if s.endswith(".hip")], []
    newest_hdr = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    for s in host:
        src, obj = os.path.join(CSRC, s), os.path.join(objdir, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), newest_hdr, os.path.getmtime(os.path.abspath(__file__))):
            continue
        cmd = ["g++"] + ASAN_FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("sanitized build failed: " + " ".join(cmd))
    if procs or not os.path.exists(ASAN_LIB) or os.path.getmtime(ASAN_LIB) < os.path.getmtime(objs[0]):
        cmd = ["g++", "-shared", "-fPIC", "-fsanitize=address,undefined", "-o", ASAN_LIB] + objs + ["-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-lpthread"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return ASAN_LIB


# ---- ThreadSanitizer lane: the host translation units through g++ -fsanitize=thread, linked with tests/tsan_precompute_driver.cpp into an
# executable (python itself would drown the report in noise) that runs the threaded host half of a first smg_precompute.
TSAN_EXE = os.path.join(LIBDIR, "tsan_precompute_driver")


def build_tsan(verbose=False):
    build(verbose=verbose)                                   # the device objects come from the regular build
    objdir = os.path.join(LIBDIR, "obj_tsan")
    os.makedirs(objdir, exist_ok=True)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    driver = os.path.join(root, "tests", "tsan_precompute_driver.cpp")
    flags = ["-std=c++17", "-O1", "-g", "-fPIC", "-fsanitize=thread", "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
             "-I" + os.path.join(root, "include"), "-w"]
    host = [s for s in SOURCES if s.endswith(".cpp")]
    objs, procs = [os.path.join(LIBDIR, "obj", os.path.splitext(s)[0] + ".o") for s in SOURCES if s.endswith(".hip")], []
    newest_hdr = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    for s in host:
        src, obj = os.path.join(CSRC, s), os.path.join(objdir, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), newest_hdr, os.path.getmtime(os.path.abspath(__file__))):
            continue
        cmd = ["g++"] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("thread-sanitized build failed: " + " ".join(cmd))
    if procs or not os.path.exists(TSAN_EXE) or os.path.getmtime(TSAN_EXE) < max(os.path.getmtime(driver), os.path.getmtime(objs[0])):
        cmd = ["g++"] + flags + [driver] + objs + ["-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-o", TSAN_EXE]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return TSAN_EXE


def sanitizer_env(base=None):
    """Environment for a python child that loads libsmg_asan.so."""
    env = dict(base if base is not None else os.environ)
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    ubsan = subprocess.check_output(["gcc", "-print-file-name=libubsan.so"], text=True).strip()
    env["LD_PRELOAD"] = asan + ":" + ubsan
    env["ASAN_OPTIONS"] = "detect_leaks=0:halt_on_error=1:abort_on_error=0:protect_shadow_gap=0:detect_odr_violation=0"
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    env["SMG_LIB"] = ASAN_LIB
    return env


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
